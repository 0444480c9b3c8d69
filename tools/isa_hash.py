"""Per-kernel hash of the gfx950 ISA of one HIP translation unit: `python tools/isa_hash.py sobfu_amd/csrc/solver_kernels.hip [out.json]`.

Compiles the file with the repo's flags and -save-temps in a scratch directory and hashes every kernel's instruction stream (labels
renumbered, comments and directives dropped).  Used to show that a source clean-up left the bench / tile instantiations' code
unchanged (profiles/r06/isa_hash_{before,after}.json); `--compare a.json b.json` lists what differs."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hashes(src):
    from sobfu_amd import build

    tmp = tempfile.mkdtemp(prefix="isa_")
    name = os.path.basename(src)
    subprocess.check_call([build._hipcc(), *build.FLAGS, *build.PER_FILE_FLAGS.get(name, []), "-save-temps", "-c", os.path.abspath(src), "-o", "x.o"], cwd=tmp,
                          stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, name.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        body = []
        for line in m.group(2).splitlines():
            line = line.split(";", 1)[0].rstrip()
            if not line or line.lstrip().startswith("."):
                if not re.match(r"\s*\.LBB\d+_\d+:", line):
                    continue
            body.append(re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", line))
        sym = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        out[re.sub(r"\(anonymous namespace\)::", "", sym)] = dict(sha256=hashlib.sha256("\n".join(body).encode()).hexdigest()[:16], instructions=len(body))
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        a, b = (json.load(open(p)) for p in sys.argv[2:4])
        for k in sorted(set(a) | set(b)):
            if k not in b:
                print("REMOVED ", k[:170])
            elif k not in a:
                print("ADDED   ", k[:170])
            elif a[k] != b[k]:
                print("CHANGED ", k[:170], a[k], b[k])
        print("%d kernels before, %d after, %d identical" % (len(a), len(b), sum(1 for k in a if k in b and a[k] == b[k])))
    else:
        h = hashes(sys.argv[1])
        json.dump(h, open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, indent=1, sort_keys=True)
