"""cProfile of the real training loop (tools/psnr_recipe.py's run): where the HOST time of a step goes once the loop is in its steady state"""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('psnr_recipe', os.path.join(ROOT, 'tools', 'psnr_recipe.py'))
pc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pc)
import torch
from arcnerf_amd import trainer as T

real = T.train_epoch
state = {'prof': None, 'n': 0}


def wrapped(*a, **k):
    epoch = a[6] if len(a) > 6 else k['epoch']
    if epoch == 2000:
        torch.cuda.synchronize()
        state['prof'] = cProfile.Profile()
        state['prof'].enable()
    r = real(*a, **k)
    if epoch == 2999 and state['prof'] is not None:
        state['prof'].disable()
    return r


T.train_epoch = wrapped
pc.run(3000, seed=0, verbose=False, report=())
st = pstats.Stats(state['prof'])
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(30)
