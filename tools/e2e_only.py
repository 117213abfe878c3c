#!/usr/bin/env python3
"""The end-to-end part of bench.py alone: N decoder processes behind the reference decoder's CLI sharing GPU 0.
usage: tools/e2e_only.py [processes [loops]]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print(json.dumps(bench.end_to_end(0, n, loops), indent=1))
