"""Per-kernel time of ONE training step from a rocprofv3 kernel trace of tools/train_graph_probe.py (rocpd database): python tools/train_kernel_table.py <results.db>"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if 'hr_features' in n]
a, b = idx[40], idx[41]
tot = collections.OrderedDict()
for n, s, e, gx, gy, gz, w in rows[a:b]:
    k = n.split('(')[0][:84]
    tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += (e - s) / 1e3
print(f'one eager step under the profiler: span {(rows[b][1] - rows[a][1]) / 1e3:.1f} us, {b - a} kernels, sum of kernel times {sum(v[1] for v in tot.values()):.1f} us')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f'{v[1]:9.1f} us  x{v[0]:3d}  {k}')
print('--- the library\'s kernels in launch order')
for n, s, e, gx, gy, gz, w in rows[a:b]:
    if 'hr_' in n:
        print(f'{(e - s) / 1e3:8.1f} us  grid {gx // max(w, 1):5d} x {gy:3d} x {gz:3d}  {n.split("(")[0][:70]}')
