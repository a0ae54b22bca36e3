"""CPU-side gate of a round's closing record (tools/round_final.sh): exit 1 unless the committed profiles/<tag>_gpu_suite.txt says the WHOLE
GPU suite was green on exactly the libdiffcloth_hip.so that is in the tree now (sha256)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(rec):
    lines = open(rec).read().splitlines()
    want = next((l.split()[1] for l in lines if l.startswith("lib_sha256 ")), None)
    rc = next((l.split()[1] for l in lines if l.startswith("suite_rc ")), None)
    lib = os.path.join(ROOT, "diffcloth_amd", "lib", "libdiffcloth_hip.so")
    have = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
    ok = want is not None and want == have and rc == "0"
    print(f"{rec}: suite rc {rc}, recorded library {str(want)[:16]}..., library in the tree {str(have)[:16]}... -> {'OK' if ok else 'REFUSED: the closing suite did not run (green) on this library'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_gpu_suite.txt")))
