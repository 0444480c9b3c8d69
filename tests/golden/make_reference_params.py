"""Extracts the parameter sets the reference ships (params/params_*.ini -- data files, key = value) into
tests/golden/reference_params.json.  Dev container only (reads /root/reference); the JSON travels with the repo.

    python tests/golden/make_reference_params.py
"""
import glob
import json
import os

REF = "/root/reference/params"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "params_*.ini"))):
        kv = {}
        for line in open(path):
            line = line.split("#", 1)[0]
            if "=" in line:
                k, v = line.split("=", 1)
                kv[k.strip()] = v.strip()
        if kv:  # params_ours.ini holds comments only
            out[os.path.basename(path)] = kv
    with open(os.path.join(HERE, "reference_params.json"), "w") as f:
        json.dump({"source": "params/params_*.ini of the reference (values as written there)", "sets": out}, f, indent=1, sort_keys=True)
        f.write("\n")
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
