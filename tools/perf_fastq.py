"""Throughput of the device FASTQ ingest: n records x L bases of synthetic FASTQ
text, uploaded once, then index + extract timed on the device (wall clock around the two
C-ABI calls, which synchronise).  Algorithmic bytes: the text is read by the two line passes
and the copy (3x), sequences + qualities + names are written once."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import _lib
from sarlacc_amd._lib import check
from sarlacc_amd.resident import DevBuffer


def synth(n, L, rng):
    name = np.frombuffer(b"@read_0000000 ch=12 start_time=1234\n", dtype=np.uint8)
    rec = np.empty((n, name.size + L + 1 + 2 + L + 1), np.uint8)
    rec[:, :name.size] = name
    idx = np.arange(n)
    for d in range(7):  # decimal read number
        rec[:, 6 + 6 - d] = 48 + (idx // 10 ** d) % 10
    o = name.size
    rec[:, o:o + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, L))]
    rec[:, o + L] = 10
    rec[:, o + L + 1] = ord("+")
    rec[:, o + L + 2] = 10
    rec[:, o + L + 3:o + 2 * L + 3] = rng.integers(33, 74, (n, L), dtype=np.uint8)
    rec[:, -1] = 10
    return rec.reshape(-1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(1000)
    text = synth(n, L, rng)
    d_text = DevBuffer.from_numpy(text)
    lib = _lib.lib()
    for rep in range(3):
        nrec, tb, tn = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        t0 = time.perf_counter()
        check(lib.sarlacc_dev_fastq_index(d_text.ptr, C.c_int64(text.size), C.byref(nrec), C.byref(tb), C.byref(tn), None))
        t1 = time.perf_counter()
        seq, qual = DevBuffer(tb.value), DevBuffer(tb.value)
        off, names, noff = DevBuffer(8 * (nrec.value + 1)), DevBuffer(tn.value), DevBuffer(8 * (nrec.value + 1))
        t2 = time.perf_counter()
        check(lib.sarlacc_dev_fastq_extract(d_text.ptr, seq.ptr, qual.ptr, off.ptr, names.ptr, noff.ptr, None))
        t3 = time.perf_counter()
        copy_ms = sarlacc_amd.last_kernel_ms()
        alg = 3 * text.size + 2 * tb.value + tn.value
        dt = (t1 - t0) + (t3 - t2)
        print("rep %d: %d records, %.2f GB text | index %.1f ms, extract %.1f ms (copy kernel %.2f ms) | "
              "%.2f GB/s of text, %.0f GB/s algorithmic traffic, %.1f M reads/s"
              % (rep, nrec.value, text.size / 1e9, (t1 - t0) * 1e3, (t3 - t2) * 1e3, copy_ms, text.size / dt / 1e9,
                 alg / dt / 1e9, nrec.value / dt / 1e6), flush=True)
    assert seq.to_numpy(np.uint8, 64).tobytes() == text[36:100].tobytes()


if __name__ == "__main__":
    main()
