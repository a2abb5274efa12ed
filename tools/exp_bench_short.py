"""the headline step only, 24 steps (for counter passes: tools/pmc_kernel.sh tools/exp_bench_short.py <kernel> "<counters>")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py', '--steps', '24', '--warmup', '4', '--no-cpu-baseline', '--no-other-configs', '--no-psnr']
os.chdir(ROOT)
import bench
bench.main()
