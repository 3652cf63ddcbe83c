"""A/B of K1 builds inside ONE lease: hr_stage_mlp over 4 x 131 072 DoNeRF rays, each library in its own process, rounds interleaved.
python tools/k1_ab.py product nopre0 nopren ...   (names of tools/_bin/libhr_<name>.so; `product` = the in-tree library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == '--child':
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import k1_operand_ubench as U
    U.child(sys.argv[2])
else:
    names = sys.argv[1:]
    res = {n: [] for n in names}
    for rnd in range(3):
        for n in names:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', n], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if r.returncode: print(n, 'rc', r.returncode, r.stderr[-400:], flush=True)
            ln = next((l for l in r.stdout.splitlines() if l.startswith('{')), None)
            if ln:
                res[n].append(json.loads(ln)['p50'])
    for n in names:
        print(n, res[n], flush=True)
