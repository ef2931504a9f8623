"""Shader clock, socket power and temperature at 10 Hz while the headline decoder runs with 2, 4 and 8 wavefronts per CU.

DESIGN 6.2 claimed (from SQ_WAVE_CYCLES / wall time) that the chip clocks down from 2.3 to 1.8 GHz when more than two of a CU's
four SIMDs run the group decoder, "and not every time"; the launch rule (runtime.hip: decode_group_lanes) rests on it.  This
tool looks at the telemetry itself: sysfs (pp_dpm_sclk, hwmon freq1_input / power1_average / power1_input / temp*_input) sampled by
a thread during every launch, and one `rocm-smi` / `amd-smi` snapshot before and after for the record.

    python tools/decode_clock_power.py [--frames 4096] [--groups 8,16,32] [--repeat 3] [--kind gradient] > profiles/r05_decode_clock_power.txt
"""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from charls_amd import batch, capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--width", type=int, default=4096)
ap.add_argument("--height", type=int, default=4096)
ap.add_argument("--groups", default="8,16,32")
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--kind", default="gradient")
ap.add_argument("--distinct", type=int, default=64, help="distinct frames (repeated): synthesis of thousands of frames takes minutes")
ap.add_argument("--hz", type=float, default=10.0)
args = ap.parse_args()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Telemetry:
    """Every sysfs source this box offers, raw: which of them exist differs between kernels."""

    def __init__(self, hz):
        self.period = 1.0 / hz
        self.sclk_files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.hwmon = {}
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
            found = sorted(glob.glob(f"/sys/class/drm/card*/device/hwmon/hwmon*/{name}"))
            if found:
                self.hwmon[name] = found[0]
        self.busy = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
        self.samples = []
        self.stop = False

    def sources(self):
        return {"pp_dpm_sclk": self.sclk_files, **self.hwmon, "gpu_busy_percent": self.busy}

    def sample(self):
        s = {"t": time.perf_counter()}
        for f in self.sclk_files[:1]:
            text = _read(f) or ""
            for line in text.splitlines():
                if "*" in line:
                    try:
                        s["sclk_mhz"] = int(line.split(":")[1].strip().lower().split("mhz")[0])
                    except (ValueError, IndexError):
                        pass
        for name, path in self.hwmon.items():
            v = _read(path)
            if v is not None and v.lstrip("-").isdigit():
                v = int(v)
                if name.startswith("freq"):
                    s[name + "_mhz"] = v // 1_000_000
                elif name.startswith("power"):
                    s[name + "_w"] = round(v / 1e6, 1)
                else:
                    s[name + "_c"] = v // 1000
        for f in self.busy[:1]:
            v = _read(f)
            if v is not None and v.isdigit():
                s["busy_pct"] = int(v)
        return s

    def _run(self):
        while not self.stop:
            self.samples.append(self.sample())
            time.sleep(self.period)

    def __enter__(self):
        self.samples, self.stop = [], False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.thread.join()

    def summary(self):
        out = {"samples": len(self.samples)}
        keys = sorted({k for s in self.samples for k in s if k != "t"})
        for k in keys:
            vals = [s[k] for s in self.samples if k in s]
            if vals:
                out[k] = {"min": min(vals), "mean": round(sum(vals) / len(vals), 1), "max": max(vals)}
        return out


def smi_snapshot(tag):
    for tool, cmd in (("amd-smi", ["amd-smi", "metric", "-c", "-p", "-t"]), ("rocm-smi", ["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"])):
        exe = shutil.which(tool) or (f"/opt/rocm/bin/{tool}" if os.path.exists(f"/opt/rocm/bin/{tool}") else None)
        if exe is None:
            continue
        try:
            r = subprocess.run([exe] + cmd[1:], capture_output=True, text=True, timeout=30)
            print(f"--- {tool} ({tag}), exit {r.returncode}")
            print("\n".join((r.stdout or r.stderr).splitlines()[:60]))
            return
        except Exception as e:  # noqa: BLE001
            print(f"--- {tool} failed: {e}")
    print(f"--- no amd-smi / rocm-smi on this box ({tag})")


lib = capi.load_product()
dev = torch.device("cuda:0")
tele = Telemetry(args.hz)
print("# decode clock / power telemetry:", json.dumps({"frames": args.frames, "kind": args.kind, "distinct": args.distinct,
                                                        "device": torch.cuda.get_device_name(0)}))
print("# telemetry sources:", json.dumps(tele.sources()))
smi_snapshot("idle, before")
n_distinct = min(args.distinct or args.frames, args.frames)
base = synth.frames_torch(n_distinct, args.width, args.height, seed0=2, bits=8, kind=args.kind, device=dev)
frames = base.repeat((args.frames + n_distinct - 1) // n_distinct, 1, 1)[:args.frames].contiguous() if n_distinct < args.frames else base
del base
out = torch.empty_like(frames)
batch.set_workspace_limit(64 << 30, lib)
enc = batch.encode_batch(frames, bits_per_sample=8, lib=lib)
torch.cuda.synchronize()
assert (enc.errcs == 0).all()
batch.release_work_areas(lib)
mpix = args.width * args.height / 1e6
with tele:
    time.sleep(1.0)
print("# idle:", json.dumps(tele.summary()))
cus = torch.cuda.get_device_properties(0).multi_processor_count
for rep in range(args.repeat):
    for g in [int(x) for x in args.groups.split(",")]:
        capi.set_knob("DECODE_GROUP", g)
        out.zero_()
        torch.cuda.synchronize()
        with tele:
            a = time.perf_counter()
            _, errcs, gpu_ms = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
            torch.cuda.synchronize()
            b = time.perf_counter()
        waves = (args.frames + 64 // g - 1) // (64 // g)
        row = {"round": rep, "lanes_per_scan": g, "wavefronts": waves, "wavefronts_per_cu": round(waves / cus, 2),
               "decode_s": round(b - a, 3), "kernel_ms": round(gpu_ms[1] if len(gpu_ms) > 1 else gpu_ms[0], 1),
               "mpix_s": round(mpix * args.frames / (b - a), 1), "ns_per_step": round((b - a) * 1e9 / (args.width * args.height), 1),
               "ok": bool((errcs == 0).all()), **tele.summary()}
        print(json.dumps(row), flush=True)
        series = [(round(s["t"] - a, 2), s.get("sclk_mhz", s.get("freq1_input_mhz")), s.get("power1_average_w", s.get("power1_input_w")))
                  for s in tele.samples]
        print("#   (t, sclk MHz, W):", " ".join(f"({t},{c},{p})" for t, c, p in series[:64]), flush=True)
capi.set_knob("DECODE_GROUP", None)
smi_snapshot("after")
