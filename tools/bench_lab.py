"""Developer tool: bench.py on the DEVELOPER library (csrc/lab/ schedules selectable through --tune), for same-box A/B runs of
the pipelined rate, e.g. `python tools/bench_lab.py --tune 1=13 --steps 12 --warmup 3 --no-cpu-baseline ...`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from tools.labenv import use_lab_library  # noqa: E402

use_lab_library()
import bench  # noqa: E402

bench.main()
