"""The ZSTD page decoder's logic, checked without a GPU: `csrc/zstd_dec.h` compiles as plain C++ with one "lane" (the wave's 64 lanes
become a loop of one); this test builds that with g++ -- with AddressSanitizer and UBSan, which the GPU pool cannot run -- and compares
it with libzstd (through pyarrow) on frames of every block and literal kind, and feeds it damaged frames.  The library itself never
decodes on the host: the parity tests proper are tests/test_gpu_scan_decode.py (`-m gpu`)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pyarrow as pa
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "arrow-ballista_amd", "csrc")

HARNESS = r'''
#include "zstd_dec.h"
#include <vector>
extern "C" int zs_host_decode(const unsigned char* in, long in_len, unsigned char* out, long out_len) {
  static gpuq::zs::Shared S;
  std::vector<unsigned char> c(in, in + in_len), o((unsigned long)out_len), scratch(gpuq::zs::BLOCK_MAX + 64);      // exact-size heap copies: an overrun is ASan's
  // every other call with the device's LDS ring and literal window (here: heap arrays of exactly their size)
  static int flip = 0; flip = (flip + 1) % 3;      // 1: ring + windows, 2: windows without the ring (what the kernel passes), 0: neither
  std::vector<unsigned char> ring(gpuq::zs::RING), litw(gpuq::zs::LITW); std::vector<uint64_t> bitw(gpuq::zs::BITW / 8 + 2), hufw(4 * (gpuq::zs::HUFW / 8 + 2));
  const bool ok = flip ? gpuq::zs::decode_frames(c.data(), in_len, o.data(), out_len, scratch.data(), S, gpuq::zs::Lds{flip == 1 ? ring.data() : nullptr, litw.data(), bitw.data(), hufw.data()})
                       : gpuq::zs::decode_frames(c.data(), in_len, o.data(), out_len, scratch.data(), S);
  for (long i = 0; i < out_len; ++i) out[i] = o[(unsigned long)i];
  return ok ? 0 : 1;
}
'''


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("zs")
    src = d / "host.cpp"
    src.write_text(HARNESS)
    so = d / "libzs_host.so"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", CSRC, str(src), "-o", str(so)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the sanitizer runtime has to be in the process before the library: run the checks in a child that preloads it
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return str(so), asan


CHILD = r'''
import ctypes, sys, numpy as np, pyarrow as pa
L = ctypes.CDLL(sys.argv[1])
L.zs_host_decode.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_long]
rng = np.random.default_rng(1)
words = [b"DELIVER IN PERSON", b"COLLECT COD", b"NONE", b"TAKE BACK RETURN"]
cases = {
    "empty": b"", "one": b"x", "zeros": bytes(300000), "noise": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
    "text": b"the quick brown fox jumps over the lazy dog " * 9000, "ascending": np.arange(150000, dtype=np.int64).tobytes(),
    "lowcard": rng.integers(0, 7, 400000, dtype=np.uint8).tobytes(), "skewed": np.minimum(rng.geometric(0.3, 500000), 255).astype(np.uint8).tobytes(),
    "int32": rng.integers(0, 5000, 300000, dtype=np.int32).tobytes(), "prices": rng.integers(90000, 10500000, 200000, dtype=np.int64).tobytes(),
    "words": b"".join(words[i] for i in rng.integers(0, 4, 60000)),
    "mixed": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes() + bytes(50000) + b"abc" * 40000,
}
for n in (1, 2, 3, 7, 63, 64, 65, 255, 256, 1000, 131071, 131072, 131073):
    cases["n%d" % n] = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
bad = []
frames = []
for name, raw in cases.items():
    for level in (1, 3, 9, 19):
        c = pa.Codec("zstd", compression_level=level).compress(raw, asbytes=True)
        out = ctypes.create_string_buffer(max(len(raw), 1))
        for variant in (0, 1, 2):      # (the harness cycles: ring + windows, windows only, neither)
            if L.zs_host_decode(c, len(c), out, len(raw)) != 0 or out.raw[:len(raw)] != raw:
                bad.append((name, level, variant))
        if len(raw) <= 300000 and level in (1, 19):
            frames.append((c, raw))
# two frames back to back, and a skippable frame in front
a, b = frames[3], frames[8]
cat = b"\x50\x2a\x4d\x18" + (5).to_bytes(4, "little") + b"hello" + a[0] + b[0]
out = ctypes.create_string_buffer(len(a[1]) + len(b[1]))
if L.zs_host_decode(cat, len(cat), out, len(out.raw)) != 0 or out.raw != a[1] + b[1]:
    bad.append(("concatenated", 0))
# damaged frames: an error or other in-bounds bytes, never an access outside the buffers (ASan / UBSan abort the process)
r = np.random.default_rng(5)
ok = rej = 0
for c, raw in frames[::3]:
    for it in range(120):
        d = bytearray(c); want = len(raw)
        kind = it % 4
        if kind == 0: d = d[:int(r.integers(0, len(d)))]
        elif kind == 1:
            for _ in range(1 + it % 3): d[int(r.integers(0, len(d)))] ^= 1 << int(r.integers(0, 8))
        elif kind == 2: d[int(r.integers(0, len(d)))] = int(r.integers(0, 256))
        else: want = int(r.integers(0, len(raw) + 2))
        out = ctypes.create_string_buffer(max(want, 1))
        if L.zs_host_decode(bytes(d), len(d), out, want) == 0: ok += 1
        else: rej += 1
print("bad", bad, "damaged ok", ok, "rejected", rej)
sys.exit(1 if bad or rej == 0 else 0)
'''


def test_host_build_of_the_decoder_equals_libzstd_and_survives_damage(host):
    so, asan = host
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", CHILD, so], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
