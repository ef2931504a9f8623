"""How many kernels of different streams does the device run side by side?  T threads decode one 1024 x 1024 stream each through
the host-pointer ABI with the coalescer OFF (every call launches its own one-scan kernel on its own stream, ~0.21 s each):
wall time ~0.21 s means they ran together, ~T x 0.21 s that they ran one after the other.  Then the same with the coalescer on.

    python tools/concurrent_kernels_probe.py > profiles/r05_concurrent_kernels.txt"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from charls_amd import capi, synth  # noqa: E402

lib = capi.load_product()
size = 1024
frames = [synth.frame_numpy(size, size, seed=50 + i) for i in range(32)]
streams = [lib.encode(f, width=size, height=size) for f in frames]
lib.decode(streams[0])


def run(threads):
    barrier = threading.Barrier(threads + 1)
    done = []

    def worker(i):
        barrier.wait()
        _, px = lib.decode(streams[i % len(streams)])
        done.append(px.tobytes() == frames[i % len(frames)].tobytes())

    pool = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for t in pool:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in pool:
        t.join()
    assert all(done)
    return time.perf_counter() - t0


for coalesce in (0, 1):
    capi.set_knob("COALESCE", coalesce)
    before = capi.engine_counters(lib)
    for threads in (1, 2, 4, 8, 16, 32):
        best = min(run(threads) for _ in range(2))
        print(f"coalescer {'on ' if coalesce else 'off'}: {threads:3d} threads x one {size}x{size} decode: {best * 1e3:7.1f} ms "
              f"({threads * size * size / 1e6 / best:7.1f} MPix/s)", flush=True)
    after = capi.engine_counters(lib)
    print(f"   engine counters of these runs: {{k: after[k] - before[k] for k in ('calls', 'launches', 'merged_calls')}}".replace(
        "{k: after[k] - before[k] for k in ('calls', 'launches', 'merged_calls')}",
        str({k: after[k] - before[k] for k in ("calls", "launches", "merged_calls")})), flush=True)
capi.set_knob("COALESCE", None)

# ---- does a decoder kernel that runs for seconds hold up an encoder call of another thread?  (It did, for as long as it ran,
# while both were launched on the handles' own streams: the device runs the kernels of few hardware queues side by side.)
big = synth.frame_numpy(4096, 4096, seed=2)
big_jls = lib.encode(big, width=4096, height=4096)
lat = []


def long_decode():
    lib.decode(big_jls)


t = threading.Thread(target=long_decode)
t0 = time.perf_counter()
t.start()
time.sleep(0.5)
for _ in range(5):
    a = time.perf_counter()
    lib.encode(big, width=4096, height=4096)
    lat.append((time.perf_counter() - a) * 1e3)
small = frames[0]
a = time.perf_counter()
lib.decode(streams[0])
small_decode_ms = (time.perf_counter() - a) * 1e3
t.join()
print(f"one 4096x4096 decode running ({time.perf_counter() - t0:.2f} s); meanwhile on another thread: 4096x4096 encodes "
      f"{', '.join(f'{v:.1f}' for v in lat)} ms, one 1024x1024 decode {small_decode_ms:.1f} ms", flush=True)
