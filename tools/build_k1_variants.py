"""Measurement variants of the f16f8 MLP kernel next to the product: only mlp_f16f8_kernel.hip is recompiled, the other objects are the product's.
python tools/build_k1_variants.py name=-DFLAG[,-DFLAG2] ...   ->  tools/_bin/libhr_<name>.so"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import build as B
B.build()
out = os.path.join(ROOT, 'tools', '_bin')
os.makedirs(out, exist_ok=True)


def one(item):
    name, fl = item
    o = os.path.join(out, f'{name}_mlp_f16f8_kernel.o')
    subprocess.run([B.hipcc(), *B.FLAGS, *fl, '-c', os.path.join(B.CSRC, 'mlp_f16f8_kernel.hip'), '-o', o], check=True, stderr=subprocess.DEVNULL)
    objs = [o if s == 'mlp_f16f8_kernel.hip' else B._obj(s) for s in B.SOURCES]
    lib = os.path.join(out, f'libhr_{name}.so')
    subprocess.run([B.hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib], check=True)
    os.remove(o)
    return lib


items = [(a.split('=')[0], [f for f in a.split('=')[1].split(',') if f]) for a in sys.argv[1:]]
with ThreadPoolExecutor(6) as ex:
    for lib in ex.map(one, items):
        print(lib)
