"""The THREADING contract of the drop-in boundary: distinct handles are independent, callers scale by threads x
handles -- the reference has no shared mutable state (src/golomb_lut.cpp:65-69, src/quantization_lut.cpp:40-62) and every
coding call builds its own codec (src/charls_jpegls_decoder.cpp:177-201).  GPU only.

T host threads, one handle per call, mixed encode / decode of DISTINCT frames (8- and 16-bit gray, sample-interleaved RGB with
HP1, a planar RGB frame, a stream with restart intervals, a near-lossless frame) through the 48 symbols only.  Every result
must equal the oracle's bytes / pixels; the HBM the library holds for its work areas (all threads together) stays within the
configured limit; no scan lands on the one-wavefront encoder for want of a work area; and concurrent calls are merged into
shared launches (charls_amd_engine_counters)."""
import threading

import numpy as np
import pytest

import oracle_bind as ob
from charls_amd import batch, capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0, "GPU box without a usable device: the product must not fall back"
    return L


def _cases(variants, lib):
    """[(name, image, encode kwargs, expected stream)]: `variants` distinct frames of every kind.  The expected stream is the
    oracle's; for the frame with restart intervals (this library's encoder extension, which the reference's encoder does not have) it is
    what ONE thread gets from this library, checked by decoding it with the oracle."""
    out = []
    for v in range(variants):
        g8 = synth.frame_numpy(320, 200, seed=100 + v, kind="mixed")
        out.append((f"gray8_{v}", g8, dict(width=320, height=200)))
        g16 = synth.frame_numpy(200, 144, seed=200 + v, bits=16, kind="mixed")
        out.append((f"gray16_{v}", g16, dict(width=200, height=144, bits_per_sample=16)))
        rgb = synth.frame_numpy(160, 120, seed=300 + v, components=3, kind="mixed")
        out.append((f"rgb_sample_hp1_{v}", rgb, dict(width=160, height=120, component_count=3, interleave_mode=2, color_transformation=1)))
        planar = synth.frame_numpy(128, 96, seed=400 + v, components=3, kind="mixed", interleaved=False)
        out.append((f"rgb_planar_{v}", planar, dict(width=128, height=96, component_count=3, interleave_mode=0)))
        near = synth.frame_numpy(256, 128, seed=500 + v, kind="hard")
        out.append((f"gray8_near2_{v}", near, dict(width=256, height=128, near_lossless=2)))
        dri = synth.frame_numpy(192, 160, seed=600 + v, kind="mixed")
        out.append((f"gray8_dri16_{v}", dri, dict(width=192, height=160, restart_interval=16)))
    done = []
    for name, img, kw in out:
        if "restart_interval" in kw:
            want = lib.encode(img, **kw)
            assert ob.decode(want)[1].tobytes() == img.tobytes()
        else:
            want = ob.encode(img, **kw)
        done.append((name, img, kw, want))
    return done


def _run_threads(lib, cases, threads, loops, align=True):
    """Every thread codes its own sequence of cases; returns the list of mismatches and the peak of work_area_bytes."""
    failures, peak = [], [0]
    barrier = threading.Barrier(threads) if align else None
    stop = threading.Event()

    def watcher():
        while not stop.is_set():
            peak[0] = max(peak[0], batch.work_area_bytes(lib))
            stop.wait(0.002)

    def worker(t):
        try:
            for k in range(loops):
                name, img, kw, want = cases[(t * 7 + k * 3) % len(cases)]
                if barrier is not None:
                    barrier.wait(timeout=120)
                got = lib.encode(img, **kw)
                if got != want:
                    failures.append((t, k, name, "encode"))
                _, px = lib.decode(want)
                expect = ob.decode(want)[1] if kw.get("near_lossless") else img
                if px.tobytes() != np.ascontiguousarray(expect).tobytes():
                    failures.append((t, k, name, "decode"))
        except BaseException as e:  # noqa: BLE001 -- reported by the test
            failures.append((t, -1, repr(e), "exception"))
            if barrier is not None:
                barrier.abort()

    w = threading.Thread(target=watcher)
    w.start()
    pool = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    stop.set()
    w.join()
    return failures, peak[0]


@pytest.mark.parametrize("threads,loops", [(8, 6), (64, 3), (256, 2)])
def test_threads_times_handles_match_the_oracle(lib, threads, loops):
    limit = 3 << 30
    batch.set_workspace_limit(limit, lib)
    try:
        cases = _cases(4, lib)
        before = capi.engine_counters(lib)
        failures, peak = _run_threads(lib, cases, threads, loops)
        after = capi.engine_counters(lib)
        assert not failures, failures[:8]
        assert peak <= limit, f"work areas grew to {peak} bytes with a limit of {limit}"
        assert after["pipeline_fallback_scans"] == before["pipeline_fallback_scans"], "a pipeline scan ran on the one-wavefront kernel"
        assert after["calls"] - before["calls"] >= threads * loops * 2
        if threads >= 64:  # (the threads are aligned by a barrier: their calls overlap)
            assert after["merged_calls"] > before["merged_calls"], (before, after)
            assert after["launches"] - before["launches"] < after["calls"] - before["calls"]
    finally:
        batch.set_workspace_limit(0, lib)
        batch.release_work_areas(lib)


def test_threads_without_the_coalescer(lib, knobs):
    """COALESCE=0: every call launches for itself, concurrently, on its own stream and its own work areas."""
    knobs.set("COALESCE", 0)
    cases = _cases(2, lib)
    before = capi.engine_counters(lib)
    failures, _ = _run_threads(lib, cases, 16, 3)
    assert not failures, failures[:8]
    assert capi.engine_counters(lib)["calls"] == before["calls"]  # (nothing went through the coalescer)
    batch.release_work_areas(lib)


def test_concurrent_decodes_share_one_launch(lib, knobs):
    """64 threads, one 1024 x 1024 stream each, released together: they must end up in a few launches (a launch alone decodes
    ONE scan per wavefront-sized kernel), and every thread gets ITS pixels."""
    knobs.set("COALESCE_WAIT_US", 2_000_000)  # (never the limiting factor here: the leader leaves when everybody announced has arrived)
    n = 64
    frames = [synth.frame_numpy(1024, 1024, seed=900 + i, kind="mixed" if i % 2 else "gradient") for i in range(n)]
    streams = [lib.encode(f, width=1024, height=1024) for f in frames]
    assert streams[5] == ob.encode(frames[5], width=1024, height=1024)
    before = capi.engine_counters(lib)
    wrong = []
    barrier = threading.Barrier(n)

    def worker(i):
        barrier.wait(timeout=60)
        _, px = lib.decode(streams[i])
        if px.tobytes() != frames[i].tobytes():
            wrong.append(i)

    pool = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    after = capi.engine_counters(lib)
    assert not wrong, wrong
    launches = after["launches"] - before["launches"]
    assert after["calls"] - before["calls"] == n and launches <= n // 4, (before, after)
    assert after["largest_launch"] >= 8


def test_mixed_sizes_and_modes_keep_apart(lib):
    """Threads coding frames of DIFFERENT geometry and mode at the same time: the coalescer must only merge what one launch can
    take (same geometry and coding parameters), and nobody waits for ever."""
    shapes = [(64, 64, 8), (65, 64, 8), (64, 65, 8), (64, 64, 12), (300, 7, 8), (1, 400, 16)]
    work = []
    for i, (w, h, bits) in enumerate(shapes * 4):
        img = synth.frame_numpy(w, h, seed=40 + i, bits=bits, kind="mixed")
        kw = dict(width=w, height=h, bits_per_sample=bits)
        work.append((img, kw, ob.encode(img, **kw)))
    wrong = []
    barrier = threading.Barrier(len(work))

    def worker(i):
        img, kw, want = work[i]
        barrier.wait(timeout=60)
        for _ in range(3):
            if lib.encode(img, **kw) != want or lib.decode(want)[1].tobytes() != img.tobytes():
                wrong.append(i)

    pool = [threading.Thread(target=worker, args=(i,)) for i in range(len(work))]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    assert not wrong, wrong


def test_failed_calls_do_not_stall_the_others(lib, knobs):
    """A call that announces itself (its upload runs) and then fails before it submits a scan -- a destination with room for
    the frame header but not for the scan header: the writer raises destination_too_small between the upload and the scan
    -- must not leave the others waiting for it: the announcement is taken back when the call ends."""
    import time
    knobs.set("COALESCE_WAIT_US", 60_000_000)  # (a leader that waited for a call that never comes would sit here for a minute)
    good = synth.frame_numpy(256, 256, seed=77, kind="mixed")
    want = ob.encode(good, width=256, height=256)
    results = []
    barrier = threading.Barrier(12)

    def worker(i):
        barrier.wait(timeout=60)
        for _ in range(3):
            if i % 3 == 0:
                try:
                    lib.encode(good, width=256, height=256, destination_size=20)  # SOI + SOF55 fit, the SOS segment does not
                    results.append(("no error", i))
                except capi.JpegLSError as e:
                    if e.errc != 3:
                        results.append((e.errc, i))
            elif lib.encode(good, width=256, height=256) != want:
                results.append(("bytes", i))

    t0 = time.perf_counter()
    pool = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
    for t in pool:
        t.start()
    for t in pool:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in pool), "a thread is still waiting"
    assert not results, results
    assert time.perf_counter() - t0 < 30, "somebody waited for a call that had failed"


def test_what_the_library_keeps_between_calls_decays(lib, knobs):
    """A drop-in library is a guest in the caller's HBM: after a burst of calls the shared work areas, the idle pool of handle
    resources and every block whose hipFree was put off are given back once no call has come for CHARLS_AMD_IDLE_RELEASE_MS
    (two seconds by default; the test asks for half a second) -- without anybody calling charls_amd_release_work_areas()."""
    import time
    knobs.set("IDLE_RELEASE_MS", 500)
    batch.release_work_areas(lib)  # (what earlier tests left in THIS thread's own areas is not the housekeeping thread's business)
    cases = _cases(2, lib)
    before = capi.engine_counters(lib)
    failures, _ = _run_threads(lib, cases, 64, 2)
    assert not failures, failures[:8]
    held_now = capi.engine_counters(lib)
    assert held_now["idle_pool_bytes"] > 0 or batch.work_area_bytes(lib) > 0, "nothing was kept: the test would prove nothing"
    deadline = time.time() + 5.0
    while time.time() < deadline:
        c = capi.engine_counters(lib)
        if (batch.work_area_bytes(lib) <= (1 << 30) and c["idle_pool_bytes"] == 0 and c["deferred_free_bytes"] == 0
                and c["idle_releases"] > before["idle_releases"]):  # (counted when the release is through, a moment after the bytes are gone)
            break
        time.sleep(0.1)
    c = capi.engine_counters(lib)
    assert batch.work_area_bytes(lib) <= (1 << 30), batch.work_area_bytes(lib)
    assert c["idle_pool_bytes"] == 0 and c["deferred_free_bytes"] == 0, c
    assert c["idle_releases"] > before["idle_releases"]
    # and the next call simply allocates again
    name, img, kw, want = cases[0]
    assert lib.encode(img, **kw) == want


def test_a_handle_that_only_reads_its_header_holds_nobody_up(lib, knobs):
    """A decoder that was given its source and read the header -- and then codes nothing -- announced a call that never comes.
    Its announcement counts for a millisecond: another thread's decode beside it takes what it takes alone (+ 1 ms at most), not
    the leader's whole wait -- set to 300 ms here, so that the difference is not a matter of clocks."""
    import time
    knobs.set("COALESCE_WAIT_US", 300_000)
    img = synth.frame_numpy(256, 256, seed=77, kind="gradient")
    stream = lib.encode(img, width=256, height=256)

    def timed_decode():
        t0 = time.perf_counter()
        _, px = lib.decode(stream)
        dt = time.perf_counter() - t0
        assert px.tobytes() == img.tobytes()
        return dt

    timed_decode()
    alone = min(timed_decode() for _ in range(5))
    beside = []
    L = lib.lib
    for _ in range(5):
        idle_handle = L.charls_jpegls_decoder_create()  # set_source_buffer + read_header, nothing else
        ptr, n, keep = lib._buf(stream)
        assert L.charls_jpegls_decoder_set_source_buffer(idle_handle, ptr, n) == 0
        assert L.charls_jpegls_decoder_read_header(idle_handle) == 0
        beside.append(timed_decode())
        L.charls_jpegls_decoder_destroy(idle_handle)
        del keep
    assert min(beside) <= alone + 0.050, (alone, beside)  # (an announcement that counted for the leader's wait would add 300 ms)
