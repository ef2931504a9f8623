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


def my_card():
    """The /sys/class/drm/cardN of the GPU this process sees (a box may expose the sysfs nodes of GPUs that belong to other
    tenants: card0 is NOT necessarily ours).  Matched by PCI address; None when that cannot be told."""
    try:
        p = torch.cuda.get_device_properties(0)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except AttributeError:
        return None, None
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(card):
            continue
        try:
            where = os.path.realpath(os.path.join(card, "device"))
        except OSError:
            continue
        if os.path.basename(where).lower().startswith(want):
            return card, want
    return None, want


class Telemetry:
    """Every sysfs source of OUR card, raw (which of them exist differs between kernels); when the card cannot be identified by
    its PCI address, all cards are sampled and the one that was busiest during the launch is reported."""

    NAMES = ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap", "temp1_input", "temp2_input", "temp3_input")

    def __init__(self, hz):
        self.period = 1.0 / hz
        self.card, self.pci = my_card()
        cards = [self.card] if self.card else [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")) if "-" not in os.path.basename(c)]
        self.cards = {}
        for c in cards:
            files = {"sclk": os.path.join(c, "device", "pp_dpm_sclk"), "busy": os.path.join(c, "device", "gpu_busy_percent")}
            for name in self.NAMES:
                found = sorted(glob.glob(os.path.join(c, "device", "hwmon", "hwmon*", name)))
                if found:
                    files[name] = found[0]
            self.cards[os.path.basename(c)] = files
        self.samples = []
        self.stop = False

    def sources(self):
        return {"my_card": self.card, "pci": self.pci, "cards": {c: sorted(f) for c, f in self.cards.items()}}

    def sample_card(self, files):
        s = {}
        text = _read(files["sclk"]) or ""
        for line in text.splitlines():
            if "*" in line:
                try:
                    s["sclk_mhz"] = int(line.split(":")[1].strip().lower().split("mhz")[0])
                except (ValueError, IndexError):
                    pass
        v = _read(files["busy"])
        if v is not None and v.isdigit():
            s["busy_pct"] = int(v)
        for name in self.NAMES:
            if name not in files:
                continue
            v = _read(files[name])
            if v is None or not v.lstrip("-").isdigit():
                continue
            v = int(v)
            if name.startswith("freq"):
                s[name + "_mhz"] = v // 1_000_000
            elif name.startswith("power"):
                s[name + "_w"] = round(v / 1e6, 1)
            else:
                s[name + "_c"] = v // 1000
        return s

    def _run(self):
        while not self.stop:
            self.samples.append({"t": time.perf_counter(), **{c: self.sample_card(f) for c, f in self.cards.items()}})
            time.sleep(self.period)

    def __enter__(self):
        self.samples, self.stop = [], False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.thread.join()

    def busiest(self):
        best, score = None, -1.0
        for c in self.cards:
            vals = [s[c].get("busy_pct", 0) for s in self.samples if c in s]
            mean = sum(vals) / len(vals) if vals else 0.0
            if mean > score:
                best, score = c, mean
        return best

    def summary(self):
        card = os.path.basename(self.card) if self.card else self.busiest()
        out = {"samples": len(self.samples), "card": card, "card_identified_by": "pci address" if self.card else "busiest during the launch"}
        keys = sorted({k for s in self.samples for k in s.get(card, {})})
        for k in keys:
            vals = [s[card][k] for s in self.samples if k in s.get(card, {})]
            if vals:
                out[k] = {"min": min(vals), "mean": round(sum(vals) / len(vals), 1), "max": max(vals)}
        return out

    def series(self, t0):
        card = os.path.basename(self.card) if self.card else self.busiest()
        return [(round(s["t"] - t0, 2), s.get(card, {}).get("sclk_mhz", s.get(card, {}).get("freq1_input_mhz")),
                 s.get(card, {}).get("power1_average_w", s.get(card, {}).get("power1_input_w")), s.get(card, {}).get("busy_pct")) for s in self.samples]


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


class SmiSeries:
    """amd-smi called back to back while a launch runs (it sees the GPU of this container, whatever sysfs shows): socket power
    and the clocks of the eight XCDs."""

    def __init__(self):
        self.exe = shutil.which("amd-smi") or ("/opt/rocm/bin/amd-smi" if os.path.exists("/opt/rocm/bin/amd-smi") else None)
        self.rows, self.stop = [], False

    def _one(self):
        r = subprocess.run([self.exe, "metric", "-c", "-p"], capture_output=True, text=True, timeout=20)
        power, clocks, section = None, [], None
        for line in r.stdout.splitlines():
            t = line.strip()
            if t.startswith("SOCKET_POWER:"):
                power = t.split(":", 1)[1].strip()
            elif t.startswith("GFX_"):
                section = "gfx"
            elif t.endswith(":") and not t.startswith("GFX_"):
                section = t if t in ("POWER:", "CLOCK:") else (None if not t.startswith("GFX") else section)
            elif t.startswith("CLK:") and section == "gfx":
                clocks.append(t.split(":", 1)[1].strip().split()[0])
                section = None
        return power, clocks

    def _run(self, t0):
        while not self.stop:
            try:
                power, clocks = self._one()
                self.rows.append((round(time.perf_counter() - t0, 2), power, clocks))
            except Exception as e:  # noqa: BLE001
                self.rows.append((round(time.perf_counter() - t0, 2), repr(e), []))
                return

    def __enter__(self):
        self.rows, self.stop = [], False
        if self.exe:
            self.thread = threading.Thread(target=self._run, args=(time.perf_counter(),), daemon=True)
            self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.exe:
            self.thread.join()


lib = capi.load_product()
dev = torch.device("cuda:0")
tele = Telemetry(args.hz)
smi = SmiSeries()
print("# decode clock / power telemetry:", json.dumps({"frames": args.frames, "kind": args.kind, "distinct": args.distinct,
                                                        "device": torch.cuda.get_device_name(0)}))
print("# telemetry sources:", json.dumps(tele.sources()))
smi_snapshot("idle, before")
n_distinct = min(args.distinct or args.frames, args.frames)
if args.kind == "tulips":  # the reference's natural test image tiled to the frame size (bench.py: data_frames)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import bench
    bench.WIDTH, bench.HEIGHT = args.width, args.height
    frames = torch.empty((args.frames, args.height, args.width), dtype=torch.uint8, device=dev)
    bench.data_frames(torch, "tulips", args.frames, dev, frames)
else:
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
        capi.set_knob("DECODE_GROUP", g if g >= 0 else None)  # (-1: the library's own rule)
        out.zero_()
        torch.cuda.synchronize()
        with tele, smi:
            a = time.perf_counter()
            _, errcs, gpu_ms = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
            torch.cuda.synchronize()
            b = time.perf_counter()
        waves = (args.frames + 64 // g - 1) // (64 // g) if g > 0 else 0
        row = {"round": rep, "lanes_per_scan": g, "wavefronts": waves, "wavefronts_per_cu": round(waves / cus, 2),
               "decode_s": round(b - a, 3), "kernel_ms": round(gpu_ms[1] if len(gpu_ms) > 1 else gpu_ms[0], 1),
               "mpix_s": round(mpix * args.frames / (b - a), 1), "ns_per_step": round((b - a) * 1e9 / (args.width * args.height), 1),
               "ok": bool((errcs == 0).all()), **tele.summary()}
        print(json.dumps(row), flush=True)
        print("#   amd-smi during the launch (t s, socket power, XCD clocks MHz):", " ".join(f"({t},{p},{'/'.join(c)})" for t, p, c in smi.rows), flush=True)
        print("#   (t s, sclk MHz, W, busy %):", " ".join(f"({t},{c},{p},{u})" for t, c, p, u in tele.series(a)[:64]), flush=True)
capi.set_knob("DECODE_GROUP", None)
smi_snapshot("after")
