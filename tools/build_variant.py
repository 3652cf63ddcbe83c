"""Builds a measurement variant of the library next to the shipped one: python tools/build_variant.py <name> [flags...]
-> tools/_bin/libhr_<name>.so (objects under tools/_bin/obj_<name>/).  Tools select it with
hyperreel_amd.lib.LIB_PATH = ... before the first load(); the product never looks there."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import build as B

def main(name, extra):
    out = os.path.join(ROOT, 'tools', '_bin')
    objd = os.path.join(out, 'obj_' + name)
    os.makedirs(objd, exist_ok=True)
    def one(s):
        o = os.path.join(objd, s.replace('.hip', '.o'))
        subprocess.run([B.hipcc(), *B.FLAGS, *extra, '-c', os.path.join(B.CSRC, s), '-o', o], check=True)
        return o
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, list(B.SOURCES)))
    lib = os.path.join(out, f'libhr_{name}.so')
    subprocess.run([B.hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib], check=True)
    print(lib)

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
