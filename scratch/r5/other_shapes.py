"""python scratch/r5/other_shapes.py <config>: per-geometry table (GCC_PROFILE_SHAPES) of one bracketed, single-stream step of one of bench.py's other configs"""
import os, sys
os.environ['GCC_PROFILE_SHAPES'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], sys.argv[1], '3']
os.environ['GCC_SERIALIZE'] = '1'
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'other_one.py'))
import torch
from gcc_amd import ops
torch.cuda.synchronize()
ops.PROFILE.start(steps=1)
g['step'](0)
ops.PROFILE.step_done()
ops.PROFILE.stop()
