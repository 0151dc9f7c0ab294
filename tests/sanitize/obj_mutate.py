"""Mutation cases for the OBJ / MTL reader (csrc/host_geometry.cpp) under AddressSanitizer + UBSan: truncations, byte flips, line edits with extreme tokens
(overflowing / negative / zero indices, nan, inf, empty fields, 70-gon faces, 70 000-character lines), CRLF and tab variants, mutated or missing .mtl files.
Test infrastructure (tests/test_host_sanitizers.py); `python tests/sanitize/obj_mutate.py SEED N HARNESS` runs a longer campaign by hand (round 6: 9 500 cases, no finding)."""
import os, random, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "assets", "Resources")
EXT = ["1e39", "-1e39", "nan", "inf", "-inf", "-0", "0", "-1", "4294967295", "4294967296", "2147483648", "-2147483649", "99999999999999999999", "", "1/", "/1", "//", "1//", "//1", "1/2/3/4", "0/0/0", "-1/-1/-1", "-99999/1/1", "1e-46", "0x10", "1.2.3", "+", "-", "e", "1e", "١"]
def mutate_text(lines):
    k = random.randrange(8)
    L = list(lines)
    if not L: return L
    for _ in range(random.randint(1, 6)):
        i = random.randrange(len(L)); t = L[i].split()
        if k == 0 and t:   # replace a token by an extreme
            j = random.randrange(len(t)); t[j] = random.choice(EXT); L[i] = " ".join(t)
        elif k == 1: del L[i]
        elif k == 2: L.insert(i, L[random.randrange(len(L))])
        elif k == 3 and t: L[i] = t[0]                       # keyword with no arguments
        elif k == 4: L[i] = L[i] + " " + " ".join(random.choice(EXT) for _ in range(random.randint(1, 40)))
        elif k == 5: L[i] = "f " + " ".join(str(random.choice([random.randint(-10, 10), random.randint(1, 10**6), 0])) + random.choice(["", "/1", "//1", "/1/1", "/", "//"]) for _ in range(random.randint(0, 70)))
        elif k == 6: L[i] = random.choice(["usemtl", "usemtl nope", "mtllib", "mtllib nope.mtl", "mtllib " + "x" * 5000, "g", "o", "s off", "vn", "vt 1", "v 1 2", "v 1 2 3 4 5 6 7", "vn 0 0 0", "#", "f", "f 1", "f 1 2", "l 1 2", "p 1"])
        elif k == 7: L[i] = "x" * random.choice([1, 300, 70000])
    return L

def make_cases(out, seed, N):
    """N mutated copies of the shipped OBJ files (+ their .mtl) under `out`, one directory each; returns the .obj paths."""
    random.seed(seed)
    seeds = [f for f in os.listdir(SRC) if f.endswith(".obj") and os.path.getsize(os.path.join(SRC, f)) < 400000]
    shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
    for n in range(N):
        s = random.choice(seeds); base = open(os.path.join(SRC, s), "rb").read()
        mode = random.randrange(5)
        d = os.path.join(out, f"c{n:05d}"); os.makedirs(d)
        mtl = os.path.join(SRC, s[:-4] + ".mtl")
        if os.path.exists(mtl):
            m = open(mtl, "rb").read()
            if random.random() < 0.4:
                ml = m.decode("latin1").split("\n"); m = "\n".join(mutate_text(ml)).encode("latin1", "replace")
            if random.random() < 0.9: open(os.path.join(d, s[:-4] + ".mtl"), "wb").write(m)
        if mode == 0: data = base[:random.randrange(len(base) + 1)]
        elif mode == 1:
            b = bytearray(base)
            for _ in range(random.randint(1, 30)): b[random.randrange(len(b))] = random.randrange(256)
            data = bytes(b)
        elif mode == 2:
            lines = base.decode("latin1").split("\n")
            # keep it small so that many line-level mutations hit the interesting parts
            if len(lines) > 400: a = random.randrange(len(lines) - 300); lines = lines[:60] + lines[a:a + 300]
            data = "\n".join(mutate_text(lines)).encode("latin1", "replace")
        elif mode == 3: data = "\n".join(mutate_text(base.decode("latin1").split("\n"))).encode("latin1", "replace")
        else: data = base.replace(b"\n", b"\r\n") if random.random() < 0.5 else base.replace(b" ", b"\t ")
        open(os.path.join(d, s), "wb").write(data)
    return sorted(os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith(".obj"))


def run_cases(harness, files, batch=40):
    """Runs the harness over the files, 40 per process; a failing batch is re-run file by file.  Returns [(file, output)] of the failures."""
    fails = []
    for i in range(0, len(files), batch):
        b = files[i:i + batch]
        if subprocess.run([harness] + b, capture_output=True, text=True, timeout=600).returncode != 0:
            for f in b:
                q = subprocess.run([harness, f], capture_output=True, text=True, timeout=600)
                if q.returncode != 0: fails.append((f, (q.stdout + q.stderr)[:2000]))
    return fails


if __name__ == "__main__":
    seed, n, harness = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    files = make_cases("/tmp/mrt_obj_cases", seed, n)
    fails = run_cases(harness, files)
    for f, o in fails[:5]: print("FAIL", f); print(o)
    print("cases", len(files), "fails", len(fails))
