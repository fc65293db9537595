import os, sys, re, subprocess
sys.path.insert(0, os.environ.get("REPO", "."))
import numpy as np
if len(sys.argv) > 1:
    from oracle import oracle as O
    from sarlacc_amd import calls
    from sarlacc_amd.mock import NUC, mutate
    L, n = 70, 2
    rng = np.random.default_rng(100 + L + n)
    truth = NUC[rng.integers(0, 4, L)]
    truth = NUC[rng.integers(0, 4, L)]
    from tests.test_gpu_msa import sim_groups
    rng = np.random.default_rng(100 + L + n)
    sim_groups(rng, 1, 1, L)
    truth = NUC[rng.integers(0, 4, L)]
    rr = [mutate(truth, rng, 0.05, 0.01).tobytes().decode() for _ in range(n)]
    P = (0, -1, -5, -1, 100)
    os.environ["ORC_MSA2_DEBUG"] = "1"; os.environ["SARLACC_MSA2_DEBUG"] = "1"
    w = O.quick_msa([list(range(1, n + 1))], rr, *P)
    g = calls.quick_msa([list(range(1, n + 1))], rr, *P)
    print("RESULT", w == g)
    print("\n".join(w[0])); print("\n".join(g[0]))
else:
    out = subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True)
    txt = out.stderr
    cur = None
    parts = {"ORC": {}, "GPU": {}}
    for line in txt.splitlines():
        m = re.match(r"(ORC|GPU) round (\d+)", line)
        if m:
            cur = (m.group(1), int(m.group(2)))
            print(line)
            continue
        m = re.match(r"\s+row (\d+) part (-?\d+) :(.*)", line)
        if m and cur:
            cols = re.findall(r"\((\d+) ", m.group(3))
            parts[cur[0]].setdefault(cur[1], []).append((int(m.group(1)), int(m.group(2)), cols))
    for rnd in sorted(parts["ORC"]):
        a, b = parts["ORC"][rnd], parts["GPU"].get(rnd, [])
        for x, y in zip(a, b):
            if x != y:
                print("round", rnd, "ORC", x, "GPU", y)
        print("round", rnd, "rows", len(a), len(b))
    print(out.stdout)
