"""cProfile of the host side of a few steps: python scratch/hostprof.py cyclegan|sagan|srgan|pix2pix"""
import cProfile
import io
import os
import pstats
import runpy
import sys

which = sys.argv[1]
sys.argv = ['bench_models.py', which] + sys.argv[2:]
os.environ['GCC_HOSTPROF'] = '1'
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_models.py'), run_name='__main__')
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
    print(s.getvalue()[:9000])
