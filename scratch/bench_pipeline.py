"""Paired-image transform chain (resize 286 bicubic, crop 256, flip, normalise): GPU pipeline vs PIL on one host core."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.data import AlignedGpuPipeline
from oracle import pipeline_oracle as P
rng = np.random.RandomState(0)
imgs = [torch.from_numpy((rng.rand(256, 512, 3) * 255).astype(np.uint8)).pin_memory() for _ in range(64)]
opt = types.SimpleNamespace(preprocess='resize_and_crop', load_size=286, crop_size=256, no_flip=False)
pipe = AlignedGpuPipeline(opt)
for _ in range(2):
    pipe.batch(imgs[:16])
torch.cuda.synchronize()
t0 = time.time()
for r in range(5):
    for i in range(0, 64, 16):
        pipe.batch(imgs[i:i + 16])
torch.cuda.synchronize()
t = (time.time() - t0) / (5 * 64)
print('GPU pipeline (host uint8 -> device A, B fp32): %.1f us/image, %.0f images/s' % (t * 1e6, 1 / t))
try:
    from PIL import Image
    arr = imgs[0].numpy()
    t0 = time.time()
    for _ in range(20):
        AB = Image.fromarray(arr)
        for box in ((0, 0, 256, 256), (256, 0, 512, 256)):
            im = AB.crop(box).resize((286, 286), Image.BICUBIC).crop((10, 10, 266, 266)).transpose(Image.FLIP_LEFT_RIGHT)
            x = (np.transpose(np.asarray(im, dtype=np.float32) / 255., (2, 0, 1)) - 0.5) / 0.5
    t = (time.time() - t0) / 20
    print('PIL + numpy, one host core: %.1f us/image, %.0f images/s' % (t * 1e6, 1 / t))
except ImportError:
    pass
