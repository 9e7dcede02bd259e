#!/usr/bin/env python3
"""Timeline of the two-stream stage-1 pipeline from a `rocprofv3 --kernel-trace` database.

    rocprofv3 --kernel-trace -d gpurun_out/x -o pipe -- python bench.py --steps 6 --warmup 1 \
        --no-cpu-baseline --no-fp32-fit --no-probes
    python tools/pipe_timeline.py gpurun_out/x/**/pipe_results.db

Answers one question: when is the EXTRACTOR's stream not running a kernel, and what is the fit doing then?
Kernels are attributed to a side by name (the trace does not keep HIP stream identities).  Per image (an image
starts at the first im2col of a group of launches) it prints the extractor's busy time, its idle time, the longest
idle gaps with their neighbours, and the fit's busy time inside the same window.
"""
import collections
import sqlite3
import sys


def side(name: str) -> str:
    if ("gemm_bf16" in name or "attention_kernel" in name or "im2col" in name or "layernorm_kernel" in name
            or "ln_cast" in name or "ln_stats" in name or "patch" in name or "pos_embed" in name):
        return "vit"
    if "fit_" in name or "adam" in name or "grid_" in name or "loss" in name:
        return "fit"
    return "other"


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:48]


def main(path: str, launches_per_image: int = 2) -> None:
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        print("no kernels in", path)
        return
    t0 = rows[0][1]
    ev = [(s - t0, e - t0, n, side(n)) for n, s, e in rows]
    vit = [x for x in ev if x[3] == "vit"]
    fit = [x for x in ev if x[3] == "fit"]
    oth = [x for x in ev if x[3] == "other"]
    print(f"{len(ev)} kernels: vit {len(vit)}, fit {len(fit)}, other {len(oth)}; span {(ev[-1][1]) / 1e6:.1f} ms")

    # images: every `launches_per_image`-th im2col starts one
    starts = [x[0] for x in vit if "im2col" in x[2]]
    img_starts = starts[::launches_per_image]
    print(f"{len(starts)} extractor launches -> {len(img_starts)} images")
    bounds = img_starts + [vit[-1][1]]
    for i in range(len(img_starts)):
        a, b = bounds[i], bounds[i + 1]
        vk = [x for x in vit if a <= x[0] < b]
        busy = sum(x[1] - x[0] for x in vk)
        last_end = max(x[1] for x in vk)
        gaps = []
        for p, q in zip(vk, vk[1:]):
            g = q[0] - p[1]
            if g > 0:
                gaps.append((g, p, q))
        idle_in = sum(g for g, _, _ in gaps)
        tail = b - last_end  # from this image's last extractor kernel to the next image's first
        fk = [x for x in fit if x[1] > a and x[0] < b]
        fbusy = sum(min(x[1], b) - max(x[0], a) for x in fk)
        ok = [x for x in oth if x[1] > a and x[0] < b]
        obusy = sum(min(x[1], b) - max(x[0], a) for x in ok)
        print(f"image {i}: window {(b - a) / 1e6:7.1f} ms | extractor busy {busy / 1e6:6.1f}, gaps inside {idle_in / 1e6:5.1f}, "
              f"idle before next image {tail / 1e6:6.1f} | fit busy {fbusy / 1e6:6.1f} ({len(fk)} kernels), other {obusy / 1e6:5.1f}")
        gaps.sort(key=lambda t: -t[0])
        for g, p, q in gaps[:3]:
            if g > 50e3:
                print(f"      gap {g / 1e3:8.1f} us after {short(p[2])} (at {(p[1] - a) / 1e6:.1f} ms) before {short(q[2])}")

    # the fit side: per image of the fit (an image's fit starts at a grid_sort after a long pause or at step 0)
    gaps = collections.Counter()
    for p, q in zip(fit, fit[1:]):
        g = q[0] - p[1]
        bucket = "<5us" if g < 5e3 else "<10us" if g < 10e3 else "<20us" if g < 20e3 else "<100us" if g < 100e3 else \
            "<1ms" if g < 1e6 else ">=1ms"
        gaps[bucket] += 1
    print("fit stream, gap between consecutive kernels:", dict(gaps))
    big = [(q[0] - p[1], p, q) for p, q in zip(fit, fit[1:]) if q[0] - p[1] >= 1e6]
    for g, p, q in big[:12]:
        print(f"      fit idle {g / 1e6:7.1f} ms after {short(p[2])} at {p[1] / 1e6:.1f} ms, next {short(q[2])}")
    # what ran between the extractor's images (the "idle before next image" windows)
    for i in range(len(img_starts) - 1):
        a = max(x[1] for x in vit if bounds[i] <= x[0] < bounds[i + 1])
        b = bounds[i + 1]
        if b - a < 1e6:
            continue
        names = collections.Counter()
        for x in ev:
            if x[1] > a and x[0] < b and x[3] != "vit":
                names[short(x[2])] += (min(x[1], b) - max(x[0], a)) / 1e6
        top = ", ".join(f"{k} {v:.1f} ms" for k, v in names.most_common(4))
        print(f"   between image {i} and {i + 1} ({(b - a) / 1e6:.1f} ms): {top}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)


def interference(path: str) -> None:
    """Per extractor kernel family: least-squares fit of  duration = a + b * (fit steps that ran inside it)  over the
    launches of one size class -- b is what one fit step costs that kernel family (us per step)."""
    import bisect
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    fit_rows = [(s, e) for n, s, e, _ in rows if "fit_rows_kernel" in n]  # one per step
    fs = [x[0] for x in fit_rows]
    fam = collections.defaultdict(list)
    for n, s, e, g in rows:
        if "gemm_bf16" in n or "attention_kernel" in n:
            i = bisect.bisect_left(fs, s)
            j = bisect.bisect_left(fs, e)
            # steps whose rows kernel STARTED and ENDED inside the launch
            k = sum(1 for t in range(i, j) if fit_rows[t][1] <= e)
            fam[(short(n)[:34], g)].append(((e - s) / 1e3, k))
    print("extractor launch duration vs fit steps completed inside it (family, grid): n, base us, us per fit step")
    for key, v in sorted(fam.items()):
        if len(v) < 8:
            continue
        n = len(v)
        mx = sum(k for _, k in v) / n
        my = sum(d for d, _ in v) / n
        sxx = sum((k - mx) ** 2 for _, k in v)
        sxy = sum((k - mx) * (d - my) for d, k in v)
        b = sxy / sxx if sxx > 0 else float("nan")
        a = my - b * mx if sxx > 0 else my
        print(f"   {key[0]:36s} grid {key[1]:7d}: n {n:4d}  mean {my:8.1f} us  steps inside {mx:6.1f}  base {a:8.1f}  slope {b:7.1f} us/step")


if __name__ == "__main__" and len(sys.argv) > 1:
    interference(sys.argv[1])
