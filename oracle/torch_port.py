"""CPU port of the reference hot path in PyTorch CPU ops (multi-threaded ATen kernels).

TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT (same rules as hyperreel_oracle.py: only
tests/ and bench.py's `cpu_baseline` leg may import it).

Why a second restatement: the numpy oracle is the parity checker (straight-line, easy to
audit) but single-threaded; the CPU baseline next to the device number should run the way the
reference itself runs on a CPU -- `F.grid_sample`, `torch.cumprod`, `torch.sort`, `addmm` on all
host cores.  This file restates the same path with exactly those ops, following the same
reference lines as the numpy oracle (see its header for the file:line list); it reuses the
oracle's setup-time constants (anchors, contraction constants, head layout) and replaces only
the per-ray arithmetic.  `tests/test_oracle_golden.py` pins it against the reference goldens too.
"""
import numpy as np
import torch
import torch.nn.functional as F

from hyperreel_oracle import C0, C1, C2, HyperReelOracle


def _act(a, x):
    y = x * float(a.inner) + float(a.shift)
    if a.type == 'sigmoid':
        y = torch.sigmoid(y)
    elif a.type == 'tanh':
        y = torch.tanh(y)
    y = y * float(a.outer)
    for w, sv in reversed(getattr(a, 'ease', [])):          # EaseValue.ease_out inside its window
        if w != 1.0:
            y = w * y + (1 - w) * sv
    return y


class TorchPort:
    def __init__(self, cfg, dataset, sd, device='cpu', iteration=None):
        self.o = HyperReelOracle(cfg, dataset, sd, iteration=iteration)          # setup-time constants only
        o = self.o
        self.dev = torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.layers = [(t(w), t(b)) for w, b in o.layers]
        self.samples = t(o.samples) if hasattr(o, 'samples') else None      # voxel_grid keeps per-axis anchors instead
        self.aabb = t(o.aabb)
        self.inv_size = t(o.inv_size)
        self.basis = t(o.basis)
        if o.video:
            self.d_a, self.d_b = [t(p)[None] for p in o.d_space], [t(p)[None] for p in o.d_time]
            self.a_a, self.a_b = [t(p)[None] for p in o.a_space], [t(p)[None] for p in o.a_time]
        else:
            self.d_a, self.d_b = [t(p)[None] for p in o.d_plane], [t(p)[None] for p in o.d_line]
            self.a_a, self.a_b = [t(p)[None] for p in o.a_plane], [t(p)[None] for p in o.a_line]

    # ---- embedding ----------------------------------------------------------------------
    def _mlp(self, x):                                      # nlf/nets/mlp.py:159-172
        o = self.o
        inp = x
        for i, (w, b) in enumerate(self.layers):
            if i in o.skips:
                x = torch.cat([inp, x], -1)
            x = F.linear(x, w, b)
            if i < o.D + 1:
                x = F.leaky_relu(x, 0.01)
        return x

    def _contract_points(self, p, c=None):                  # nlf/contract.py:178-192 (mipnerf), :84-85 / :110-111 (affine)
        c = self.o.contract if c is None else c
        if hasattr(c, 'power'):                              # DoNeRFContract, contract.py:238-240
            d = torch.norm(p, dim=-1, keepdim=True)
            return (p / d) * torch.pow(d * c.fac + 1e-8, 1.0 / c.power)
        if hasattr(c, 'bbox_min'):
            lo = torch.from_numpy(c.bbox_min).to(self.dev)
            return (p - lo) / (torch.from_numpy(c.bbox_max).to(self.dev) - lo)
        if not hasattr(c, 'r0'):
            return p / float(c.fac)
        p = p / c.r0
        d = torch.norm(p, dim=-1, keepdim=True)
        inv_end = c.r0 / c.r1
        t = (1.0 / d.abs() - inv_end) * (1.0 / (1.0 - inv_end))
        return torch.where(d < 1, p, (p / d) * (2.0 - t))

    def _inv_contract_distance(self, z, c=None):            # nlf/contract.py:143-158 (mipnerf), :78-79 / :104-105 (affine)
        c = self.o.contract if c is None else c
        if hasattr(c, 'power'):                              # DoNeRFContract, contract.py:226-230
            z = ((z / 2.0) * 2.0).clamp(-2.0, 2.0)
            return torch.pow(torch.abs(z) + 1e-8, c.power) * torch.sign(z) / c.fac
        if not hasattr(c, 'r0'):
            return z * float(c.fac) if hasattr(c, 'fac') else z
        inv_end = c.d0 / c.d1
        z = (z / 2.0) * 2.0
        z = z.clamp(-2.0, 2.0)
        inv = (2.0 - z.abs()) / (1.0 / (1.0 - inv_end)) + inv_end
        return torch.where(z.abs() < 1, z, torch.sign(z) * (1.0 / inv)) * c.d0

    def _param_pe(self, rays, params=None):                 # nlf/param.py:87-115,244-253; nlf/pe.py:210-221,53-66
        cols = []
        for pcfg in (self.o.pred_cfg['params'] if params is None else params).values():
            x = rays[:, pcfg['start']:pcfg['end']]
            p = pcfg['param']
            if p['fn'] == 'pluecker':
                org = torch.tensor(p.get('origin', [0.0, 0.0, 0.0]), device=self.dev)
                d = F.normalize(x[:, 3:6], p=2.0, dim=-1)
                m = torch.cross(x[:, :3] - org[None], d, dim=-1)
                y = torch.cat([d * p.get('direction_multiplier', 1.0), m * p.get('moment_multiplier', 1.0)], -1)
            elif p['fn'] == 'two_plane':
                org = torch.tensor(p.get('origin', [0.0, 0.0, 0.0]), device=self.dev)
                oo, dd = x[:, :3] - org[None], x[:, 3:6]
                dz = torch.where(dd.abs() < 1e-5, torch.full_like(dd, 1e12), dd)[:, 2]
                t1 = (p.get('near', -1.0) - oo[:, 2]) / dz
                t2 = (p.get('far', 0.0) - oo[:, 2]) / dz
                y = torch.cat([oo[:, :2] + dd[:, :2] * t1[:, None], oo[:, :2] + dd[:, :2] * t2[:, None]], -1)
            else:
                y = x
            pe = pcfg.get('pe')
            if pe is not None and pe['n_freqs'] > 0:
                n = int(pe['n_freqs'])
                freqs = [float(pe.get('freq_multiplier', 2.0)) ** (j + 1) for j in range(n)]
                if pe['type'] == 'windowed':
                    bm = float(pe.get('base_multiplier', 1.0))
                    out = [] if pe.get('exclude_identity', False) else [y]
                    from hyperreel_oracle import windowed_pe_weights
                    for f, w in zip(freqs, windowed_pe_weights(pe, self.o.iteration)):
                        out += [w * torch.sin(bm * f * y), w * torch.cos(bm * f * y)] if w != 1.0 else \
                               [torch.sin(bm * f * y), torch.cos(bm * f * y)]
                else:
                    cur = (torch.tensor(freqs, device=self.dev)[None, None] * y[..., None]).reshape(y.shape[0], -1)
                    out = [y, torch.sin(cur), torch.cos(cur)]
                y = torch.cat(out, -1)
            cols.append(y)
        return torch.cat(cols, -1)

    def embed(self, rays, head=None, point_head=None):
        """`head`: optional (B, Z*P) raw output of the ray MLP to use instead of running it; `point_head`: the same for the
        point MLP of a cascade, (B*Zc, M*P) (gradient checks of the training path differentiate with respect to them)."""
        import hyperreel_oracle as H
        from hyperreel_oracle import Act
        o = self.o
        H.ITERATION = o.iteration                           # stages build their activations while they run
        B = rays.shape[0]
        x = {}
        for idx, typ, ecfg in o.stages:
            if typ == 'ray_prediction':                     # ray.py:316-347
                Z = o.Z
                if head is not None:
                    h = head
                elif o.zero_net:                            # ZeroMLP, nlf/nets/mlp.py:14-33
                    h = torch.zeros(B, Z * sum(o.out_shapes), device=self.dev)
                else:
                    h = self._mlp(self._param_pe(rays))
                h = h.view(B, Z, -1)
                off = 0
                for name, n, act in zip(o.out_names, o.out_shapes, o.out_acts):
                    x[name] = _act(act, h[..., off:off + n])
                    off += n
            elif typ == 'ray_intersect':
                self._intersect(o._isects[idx], rays, x)
            elif typ == 'point_prediction':                 # point.py:137-203
                pp = o._pp[idx]
                Zi = x['points'].shape[1]
                cols = []
                for name, n in pp['cfg']['inputs'].items():
                    if name == 'viewdirs':
                        cols.append(rays[:, None, 3:6].expand(B, Zi, 3))
                    elif name == 'origins':
                        cols.append(rays[:, None, 0:3].expand(B, Zi, 3))
                    elif name == 'times':
                        cols.append(rays[:, None, -1:].expand(B, Zi, 1))
                    else:
                        cols.append(x[name][..., :int(n)])
                inp = torch.cat(cols, -1).reshape(B * Zi, -1)
                x['_rows'] = inp
                if point_head is not None:
                    h = point_head
                else:
                    layers = [(torch.from_numpy(w).to(self.dev), torch.from_numpy(b).to(self.dev)) for w, b in pp['layers']]
                    h = self._run_layers(self._param_pe(inp, pp['cfg']['params']), layers, pp['skips'], pp['D'])
                h = h.view(B, -1, sum(pp['shapes']))
                off = 0
                for name, n, act in zip(pp['names'], pp['shapes'], pp['acts']):
                    x[name] = _act(act, h[..., off:off + n])
                    off += n
            elif typ == 'color_transform':                  # point.py:585-596: a no-op unless dataset.val_all
                if o.ds.get('val_all', False):
                    if getattr(self, 'color_table', None) is None:
                        self.color_table = torch.from_numpy(o.sd[f'{o.EMB}{idx}.color_embedding']).to(self.dev)
                    row = self.color_table[torch.round(rays[:, -2]).long()]
                    x['color_transform_global'] = _act(Act(ecfg.get('transform_activation')), row[:, :9])
                    x['color_shift_global'] = _act(Act(ecfg.get('shift_activation')), row[:, -3:])
            elif typ == 'advect_points':                      # point.py:780-831, flow_utils.py:10-35
                t = rays[:, -1:]
                K, Fr = o.ds['num_keyframes'], o.ds['num_frames']
                fac = K * (Fr - 1) / Fr
                base = torch.round((t * fac).clamp(0.0, K - 1.0) - 1e-5) * (1.0 / fac)
                if ecfg.get('use_spatial_flow', False):
                    x['points'] = x['points'] + _act(Act(ecfg.get('spatial_flow_activation')), x['spatial_flow']) * (t - base)[:, None, :]
                x['base_times'] = base[:, None, :].expand(B, x['points'].shape[1], 1)
            elif typ == 'point_offset':                     # point.py:371-396
                fld = ecfg.get('in_density_field', 'sigma')
                sg = x[fld] if (ecfg.get('use_sigma', True) and fld in x) else torch.zeros(B, x['points'].shape[1], 1, device=self.dev)
                x['points'] = x['points'] + _act(Act(ecfg.get('activation')), x['point_offset']) * (1 - sg)
        Z = x['points'].shape[1]
        x['viewdirs'] = rays[:, None, 3:6].expand(B, Z, 3)
        return x

    def _run_layers(self, x, layers, skips, D):             # nlf/nets/mlp.py:159-172
        inp = x
        for i, (w, b) in enumerate(layers):
            if i in skips:
                x = torch.cat([inp, x], -1)
            x = F.linear(x, w, b)
            if i < D + 1:
                x = F.leaky_relu(x, 0.01)
        return x

    def _intersect(self, o, rays, x):                       # Intersect.forward, nlf/intersect/base.py:142-259
        """`o`: the stage's _Isect (anchors, scales, contraction ... of hyperreel_oracle)."""
        B, Z = rays.shape[0], o.Z
        r = torch.cat([rays[:, :3] - torch.from_numpy(o.origin).to(self.dev)[None], rays[:, 3:6]], -1)
        sigma = x[o.in_density_field].reshape(B, -1) if (o.use_sigma and o.in_density_field in x) else torch.zeros(B, Z, device=self.dev)
        zv = _act(o.z_act, x['z_vals'].reshape(B, Z, -1)) * (1 - sigma[..., None])

        def proc(z):                                        # intersect/base.py:128-140
            z = z * float(o.z_scale) + torch.from_numpy(o.samples).to(self.dev)[None]
            return self._inv_contract_distance(z, o.contract) if o.contract.contract_samples else z

        if o.isect_type == 'euclidean_distance_unified':    # primitive.py:162-176, param.py:297-307
            z = proc(zv.reshape(B, Z))
            dn = F.normalize(r[:, 3:6], p=2.0, dim=-1)
            pos = torch.cross(dn, torch.cross(r[:, :3], dn, dim=-1), dim=-1)
            diff = pos - r[:, :3]
            dists = z + (torch.sign((r[:, 3:6] * diff).sum(-1)) * torch.norm(diff, dim=-1))[:, None]
        elif o.isect_type == 'voxel_grid':                  # voxel.py:72-112, intersect_utils.py:152-179
            nz = Z // 3
            z = zv.reshape(B, nz, 3) * torch.from_numpy(o.voxel_scale).to(self.dev)[None, None] \
                + torch.from_numpy(o.voxel_samples).to(self.dev)[None]
            if o.contract.contract_samples:
                z = self._inv_contract_distance(z, o.contract)
            if o.outward_facing:
                z = z * torch.sign(r[:, None, 3:6])
            d = r[:, None, 3:6]
            d = torch.where(d.abs() < 1e-5, torch.full_like(d, 1e12), d)
            dists = ((z - r[:, None, 0:3]) / d).reshape(B, Z)
        elif o.isect_type == 'z_plane':                     # z.py:77-97, intersect_utils.py:127-150
            z = proc(zv.reshape(B, Z))
            d = r[:, None, 3:6]
            d = torch.where(d.abs() < 1e-5, torch.full_like(d, 1e12), d)
            dists = (z - r[:, None, 2]) / d[..., 2]
        elif o.isect_type in ('sphere_new', 'cylinder_new'):     # primitive.py:305-363, 490-545
            def nrm(v):
                return torch.norm(v, dim=-1)

            def ppos(oo, dd):                                     # param.py:297-307
                dn_ = F.normalize(dd, p=2.0, dim=-1)
                return torch.cross(dn_, torch.cross(oo, dn_, dim=-1), dim=-1)

            def quad(oo, dd, radii):
                a = (dd * dd).sum(-1)
                b = 2 * (oo * dd).sum(-1)
                c = (oo * oo).sum(-1) - radii * radii
                disc = b * b - 4 * a * c
                disc = torch.where(disc < 0, torch.zeros_like(disc), disc)
                sq = torch.sqrt(disc + 1e-8)
                t1, t2 = (-b + sq) / (2 * a), (-b - sq) / (2 * a)
                t1 = torch.where(disc <= 0, torch.zeros_like(t1), t1)
                t2 = torch.where(disc <= 0, torch.zeros_like(t2), t2)
                return torch.where((t2 < 0) | (radii < 0), t1, t2)

            tn = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(self.dev)
            origins = zv[..., :3] * float(o.origin_scale)
            resize = zv[..., 3:6] * float(o.resize_scale) + tn(o.resize_initial)[None, None]
            raw_offsets, radii = proc(zv[..., 6]), proc(zv[..., 7])
            ro = (r[:, None, 0:3] - origins) * resize
            rd = r[:, None, 3:6] * resize
            rn = F.normalize(rd, p=2.0, dim=-1)
            if o.isect_type == 'sphere_new':
                tt = quad(ro, rn, radii)
                base_pos = ppos(ro, rn)
                min_radius = nrm(base_pos)
                diff = base_pos - ro
                base_distance = torch.sign((rn * diff).sum(-1)) * nrm(diff)
            else:
                xz = lambda v: torch.stack([v[..., 0], torch.zeros_like(v[..., 1]), v[..., 2]], -1)
                tt = quad(ro[..., [0, 2]], rn[..., [0, 2]], radii)
                o_c, d_c = xz(ro), xz(rn)
                base_pos = ppos(o_c, d_c)
                min_radius = nrm(base_pos)
                diff = base_pos - o_c
                base_distance = torch.sign((d_c * diff).sum(-1)) * nrm(diff) / nrm(d_c)
            recycle = radii.abs() < min_radius + 4.0 * float(o.z_scale)
            tt = torch.where(recycle, raw_offsets + base_distance, tt)
            dists = tt / (nrm(rd) + 1e-5)
        elif o.isect_type == 'deformable_voxel_grid':            # voxel.py:178-213, intersect_utils.py:210-236
            normals0 = torch.from_numpy(o.dvg_normals).to(self.dev)
            na = normals0.shape[0]
            dpl = proc(zv[..., 3])
            normal = zv[..., :3].reshape(-1, na, 3) * float(o.dvg_scale) + normals0[None]
            normal = F.normalize(normal.reshape(B, -1, 3), p=2.0, dim=-1)
            o_n = (r[:, None, :3] * normal).sum(-1)
            d_n = (r[:, None, 3:6] * normal).sum(-1)
            d_n = torch.where(d_n.abs() < 1e-5, torch.full_like(d_n, 1e12), d_n)
            dists = (dpl - o_n) / d_n
        elif o.isect_type not in ('sphere', 'cylinder'):
            raise NotImplementedError(o.isect_type)
        else:                                               # primitive.py:420-438 / 235-253
            origins = zv[..., :3] * float(o.origin_scale) + torch.from_numpy(o.origin_initial).to(self.dev)[None, None]
            radii = proc(zv[..., 3])
            oo, dd = r[:, None, 0:3] * origins, r[:, None, 3:6] * origins
            if o.isect_type == 'cylinder':
                oo, dd = oo[..., [0, 2]], dd[..., [0, 2]]
            a = (dd * dd).sum(-1)
            b = 2 * (oo * dd).sum(-1)
            c = (oo * oo).sum(-1) - radii * radii
            disc = b * b - 4 * a * c
            disc = torch.where(disc < 0, torch.zeros_like(disc), disc)
            sq = torch.sqrt(disc + 1e-8)
            t1, t2 = (-b + sq) / (2 * a), (-b - sq) / (2 * a)
            t1 = torch.where(disc <= 0, torch.zeros_like(t1), t1)
            t2 = torch.where(disc <= 0, torch.zeros_like(t2), t2)
            dists = torch.where((t2 < 0) | (radii < 0), t1, t2)
        if o.mask_on:
            mask = (dists <= float(o.near)) | (dists >= float(o.far))
            dists = torch.where(mask, torch.zeros_like(dists), dists)
        if o.sort:
            dists = torch.sort(dists, dim=1)[0]
        dists = dists[..., None]
        mask = dists == 0
        points = r[:, None, :3] + r[:, None, 3:6] * dists
        if hasattr(o.contract, 'contract_points'):          # contract.py:43-50
            oc = self._contract_points(r[:, :3], o.contract)
            points = self._contract_points(points, o.contract)
            dists = torch.norm(points - oc[:, None], dim=-1, keepdim=True)
        dists = torch.where(mask, torch.zeros_like(dists), dists)
        x['points'], x['distances'] = points, dists

    # ---- colour -------------------------------------------------------------------------
    def _feat(self, planes_a, planes_b, pn):                # tensorf_no_sample.py:47-126, tensorf_dynamic.py:287-371
        o = self.o
        out = []
        N = pn.shape[0]
        for i in range(3):
            if o.video and self.d_a[i].shape[1] == 0:
                continue
            ga = pn[:, o.MAT[i]].view(1, N, 1, 2)
            pa = F.grid_sample(planes_a[i], ga, align_corners=True).view(-1, N)
            if o.video:
                gb = pn[:, o.MAT_T[i]].view(1, N, 1, 2)
            else:
                gb = torch.stack([torch.zeros(N, device=self.dev), pn[:, o.VEC[i]]], -1).view(1, N, 1, 2)
            pb = F.grid_sample(planes_b[i], gb, align_corners=True).view(-1, N)
            out.append(pa * pb)
        return torch.cat(out, 0)

    def color(self, x, train=False, white_bg=None):
        """train=True: no eval-mode clamp (tensorf_no_sample.py:246); white_bg overrides the configured background
        (the reference draws it at random per training step, :236)."""
        o = self.o
        pts = x['points']
        B, Z = pts.shape[:2]
        dist = x['distances'].reshape(B, Z)
        deltas = torch.cat([dist[:, 1:] - dist[:, :-1], torch.full((B, 1), 1e10, device=self.dev)], 1)
        valid = ~(((self.aabb[0] > pts) | (pts > self.aabb[1])).any(-1)) & (dist > 0)
        pn = (pts - self.aabb[0]) * self.inv_size - 1
        if o.video:
            pn = torch.cat([pn, (x['base_times'] * o.tsf + o.tpo) * 2 - 1], -1)
        sigma = torch.zeros(B, Z, device=self.dev)
        if valid.any():
            f = self._feat(self.d_a, self.d_b, pn[valid]).sum(0)
            sigma[valid] = F.relu(f) if o.act == 'relu' else (f.abs() if o.act == 'relu_abs' else F.softplus(f + float(o.density_shift)))
        alpha = 1.0 - torch.exp(-sigma * (deltas * float(o.distance_scale)))
        T = torch.cumprod(torch.cat([torch.ones(B, 1, device=self.dev), 1.0 - alpha + 1e-10], -1), -1)
        weight = alpha * T[:, :-1]
        app = weight > float(o.thr)
        rgb = torch.zeros(B, Z, 3, device=self.dev)
        if app.any():
            feat = F.linear(self._feat(self.a_a, self.a_b, pn[app]).T, self.basis)
            if o.shading == 'RGB':
                col = torch.sigmoid(feat)
            else:
                d = x['viewdirs'][app]
                xx, yy, zz = d[:, 0], d[:, 1], d[:, 2]
                sh = torch.stack([torch.full_like(xx, C0), -C1 * yy, C1 * zz, -C1 * xx, C2[0] * xx * yy, C2[1] * yy * zz,
                                  C2[2] * (2.0 * zz * zz - xx * xx - yy * yy), C2[3] * xx * zz, C2[4] * (xx * xx - yy * yy)], -1)
                col = torch.relu((sh[:, None] * feat.view(-1, 3, 9)).sum(-1) + 0.5)
            rgb[app] = col
        if 'color_scale' in x:
            rgb = rgb * (x['color_scale'] + 1.0) + x['color_shift']
        out = (weight[..., None] * rgb).sum(-2)
        if o.white_bg if white_bg is None else white_bg:
            out = out + (1.0 - weight.sum(-1)[:, None])
        if 'color_scale_global' in x:                       # scale_shift_color_one, tensorf_utils.py:275-281
            out = out * (x['color_scale_global'][:, 0, :] + 1.0) + x['color_shift_global'][:, 0, :]
        elif 'color_transform_global' in x:                 # transform_color_one, tensorf_utils.py:308-320
            T = x['color_transform_global'].reshape(B, -1, 3, 3)[:, 0]        # a per-camera table row, or sample 0's nine head values
            out = out + (T * out[:, None, :]).sum(-1) + x['color_shift_global'].reshape(B, -1, 3)[:, 0]
        return out if train else out.clamp(0, 1)

    @torch.no_grad()
    def render(self, rays, chunk=16384):
        if not torch.is_tensor(rays):
            rays = torch.from_numpy(np.ascontiguousarray(rays, np.float32))
        rays = rays.to(self.dev)
        outs = [self.color(self.embed(rays[i:i + chunk])) for i in range(0, rays.shape[0], chunk)]
        out = torch.cat(outs, 0)
        return {'rgb': out.cpu().numpy() if self.dev.type == 'cpu' else out}
