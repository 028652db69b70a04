"""Patch-feed launches at training size (GPU box): stage-2 batches (image / label / uncertainty patches of
TrainSetMultipleSegSREfficient) and stage-1 (LR, HR) pairs of TrainSetMultiple, with the bytes each launch moves.

    python tools/bench_feed.py"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rehrseg_amd.utils.train_set import TrainSetMultiple, TrainSetMultipleSegSREfficient  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    rng = np.random.RandomState(0)
    shape = (400, 400, 160)
    vols = [dict(img=rng.rand(*shape).astype(np.float32), seg=(rng.rand(*shape) > 0.5).astype(np.uint8),
                 uncertainty=rng.randint(0, 256, size=shape).astype(np.uint8)) for _ in range(2)]
    for B, ps, sep in ((2, (128, 128, 32), 4), (2, (192, 192, 40), 4), (8, (128, 128, 32), 4)):
        ds = TrainSetMultipleSegSREfficient(None, [0, 1], float(sep), 1.0, ps, None, True, True, device="cuda:0", volumes=vols)
        random.seed(0)
        t = timed(lambda: ds.batch([k % 2 for k in range(B)]))
        lr, hr = ps[0] * ps[1] * ps[2], ps[0] * ps[1] * ps[2] * sep
        byts = B * (lr * (4 + 4) + lr * (1 + 4) + hr * (1 + 4) + lr * (1 + 4))  # read + write of the four outputs
        print(f"stage-2 batch {B} x {ps} sep {sep}: {t * 1e6:8.1f} us  {byts / t / 1e9:7.1f} GB/s algorithmic "
              f"(4 launches, host descriptor work included)", flush=True)
    shape = (320, 320, 96)
    image = np.stack((rng.rand(*shape).astype(np.float32), (rng.rand(*shape) > 0.5).astype(np.float32)), -1)
    for B, ps in ((1, (128, 128, 128)), (4, (128, 128, 128))):
        ds = TrainSetMultiple(None, [0], 4.0, 1.0, None, None, ps, True, "cuda:0", volumes=[image],
                              blur_kernel=np.array([0.05, 0.2, 0.5, 0.2, 0.05], np.float32))
        random.seed(0)
        t = timed(lambda: ds.batch([0] * B))
        print(f"stage-1 batch {B} x {ps}: {t * 1e6:8.1f} us (6 launches + 2 cats)", flush=True)


if __name__ == "__main__":
    main()
