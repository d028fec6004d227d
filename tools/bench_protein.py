"""Kernel rates of the protein / dayhoff / hp / translate sketches on resident input (bench.py: protein_extras, on its own):
python tools/bench_protein.py  -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from sourmash_amd import device as smd
    extra = {}
    bench.protein_extras(extra, torch, np, torch.device("cuda", 0), smd, None)
    print(json.dumps(extra))


if __name__ == "__main__":
    main()
