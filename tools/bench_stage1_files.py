"""Developer tool: BASELINE configs[3] in miniature -- the stage-1 sweep on IMAGE FILES, everything included:
JPEG decode + PIL resize on the host, on-GPU view synthesis, extractor, fit, and the two .npy writes per image
(what bench.py's timed region leaves out by SURVEY's definition).  VOC-sized synthetic JPEGs (500 x 375)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import stage1, views as V  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=12)
ap.add_argument("--num_iters", type=int, default=1000)
a = ap.parse_args()
root = tempfile.mkdtemp(prefix="dvt_files_")
os.makedirs(f"{root}/data/JPEGImages")
rng = np.random.default_rng(0)
names = []
for i in range(a.images):
    low = rng.integers(0, 255, (12, 16, 3), dtype=np.uint8)
    img = Image.fromarray(low).resize((500, 375), Image.BICUBIC)
    img.save(f"{root}/data/JPEGImages/{i:06d}.jpg", quality=92)
    names.append(f"JPEGImages/{i:06d}.jpg")
with open(f"{root}/list.txt", "w") as f:
    f.write("".join(n + "\n" for n in names))
args = stage1.get_args(["--img_path", f"{root}/list.txt", "--data_root", f"{root}/data/", "--save_root", f"{root}/out",
                        "--num_imgs", str(a.images), "--num_iters", str(a.num_iters), "--warmup_iters",
                        str(a.num_iters // 10), "--dtype", "bfloat16", "--allow_random_vit", "--output_dir", f"{root}/work"])
dev = torch.device("cuda:0")
# view synthesis alone
img = V.load_image(f"{root}/data/{names[0]}", args.input_size, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), dev)
boxes, coords = V.sample_view_boxes(768, args.input_size, 37, 37, rng=np.random.RandomState(0))
out = torch.empty(769, 3, 518, 518, device=dev)
V.render_views(img, boxes, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    V.render_views(img, boxes, out)
torch.cuda.synchronize()
t_render = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    V.load_image(f"{root}/data/{names[0]}", args.input_size, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), dev)
torch.cuda.synchronize()
t_load = (time.perf_counter() - t0) / 5
del out
t0 = time.perf_counter()
stage1.main(args, 0, 1)
torch.cuda.synchronize()
t = time.perf_counter() - t0
n_out = sum(len(fs) for _, _, fs in os.walk(f"{root}/out"))
print(f"files: {a.images} JPEG images end to end in {t:.2f} s (includes library load + first-launch warm-up) = "
      f"{a.images / t:.2f} images/s; outputs written: {n_out} .npy; render_views 769 views {t_render * 1e3:.2f} ms "
      f"({2.476e9 / t_render / 1e12:.2f} TB/s of view writes); decode + PIL resize + upload {t_load * 1e3:.1f} ms")
