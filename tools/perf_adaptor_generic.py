"""Generic-level adaptorAlign (SURVEY 8d ii): FASTQ file -> device parse -> front/back windows
(tolerance 250) -> four alignments per read with traceback and sections -> strand resolution,
on n synthetic 2-kb reads written to a temporary FASTQ file.  Cells = 4 x windows x adaptor."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import sarlacc_amd
from sarlacc_amd import generics as G
from perf_fastq import synth

A1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"
A2 = "CACACTGAGCAGCGACTAGACA"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    text = synth(n, L, np.random.default_rng(1000))
    with tempfile.NamedTemporaryFile(suffix=".fastq", delete=False, dir=os.environ.get("TMPDIR", "/tmp")) as fh:
        path = fh.name
    text.tofile(path)
    try:
        for rep in range(2):
            t0 = time.perf_counter()
            out = G.adaptorAlign(A1, A2, path, tolerance=250)
            dt = time.perf_counter() - t0
            cells = 2 * 250 * (len(A1) + len(A2)) * n
            print("rep %d: adaptorAlign(filepath) on %d reads x %d bp: %.2f s = %.2f M reads/min, %.1f GCUPS generic-level "
                  "(%d reversed)" % (rep, n, L, dt, n / dt * 60 / 1e6, cells / dt / 1e9, int(out["reversed"].sum())), flush=True)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
