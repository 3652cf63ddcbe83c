"""Host-side mirror of the reference's model objects for the forward-render path.

`HipLightfieldModel` stands where `LightfieldModel` (nlf/models/models.py:104-138) stands
in the reference: it is built from the same `experiment.model` YAML group and the same
`system` object, exposes the attributes `INRSystem` touches (`set_iter`,
`embedding_model`, `color_model.net`, nlf/__init__.py:449-479,608-614) and owns
nn.Parameters under exactly the reference's state_dict keys, so
reference checkpoints load by name.  The modules hold weights only -- all arithmetic
happens in libhyperreel_hip.so, which receives the tensors through hr_model_upload.
"""
import math

import torch
from torch import nn

from . import lib as _lib
from .config import n_to_reso, to_plain
from .plan import compile_model, hr_fields, upload_names

MAT_MODE = [[0, 1], [0, 2], [1, 2]]
VEC_MODE = [2, 1, 0]
MAT_MODE_TIME = [[2, 3], [1, 3], [0, 3]]


def dataset_scalars_from_system(system):
    """The five `system.dm.train_dataset` reads of the hot path."""
    td = system.dm.train_dataset
    get = (lambda k, d=None: td[k] if isinstance(td, dict) and k in td else getattr(td, k, d))
    ds = {'near': get('near', 0.0), 'far': get('far', 1.0), 'depth_range': list(get('depth_range', [0.0, 1.0])),
          'num_keyframes': get('num_keyframes', 1), 'num_frames': get('num_frames', 1)}
    # read by single stages only: voxel_grid bounds (voxel.py:27-29), the per-camera colour table (point.py:576-577)
    for k in ('bbox_min', 'bbox_max', 'total_images_per_frame', 'val_all'):
        v = get(k)
        if v is not None:
            ds[k] = [float(t) for t in v] if k.startswith('bbox') else v
    return ds


class _Dummy(nn.Module):
    """RayParam / PE modules of the reference carry an unused nn.Linear(1,1) `dummy_layer`
    (nlf/param.py:479, nlf/pe.py:165); kept so strict state_dict loading works."""

    def __init__(self):
        super().__init__()
        self.dummy_layer = nn.Linear(1, 1)


class _Empty(nn.Module):
    pass


class _ZeroNet(nn.Module):
    """ZeroMLP (nlf/nets/mlp.py:14-33): an unused nn.Linear(1, 1), kept for strict state_dict loading."""

    def __init__(self):
        super().__init__()
        self.layer = nn.Linear(1, 1)


class HostMLP(nn.Module):
    """Parameter container shaped like BaseMLP (nlf/nets/mlp.py:127-154)."""

    def __init__(self, shapes):
        super().__init__()
        self.layers = nn.ModuleList()
        for i, (o, n_in) in enumerate(shapes):
            lin = nn.Linear(n_in, o)
            self.layers.append(nn.Sequential(lin, nn.Identity()) if i < len(shapes) - 1 else lin)


class HostRayPrediction(nn.Module):
    def __init__(self, pred_cfg, shapes):
        super().__init__()
        self.params = nn.ModuleList([_Dummy() for _ in pred_cfg['params']])
        self.pes = nn.ModuleList([_Dummy() if 'pe' in p else _Empty() for p in pred_cfg['params'].values()])
        self.net = HostMLP(shapes) if shapes else _ZeroNet()


class HostColorTransform(nn.Module):
    """ColorTransformEmbedding (nlf/embedding/point.py:558-602): one [3x3 | shift] row per camera."""

    def __init__(self, num_views):
        super().__init__()
        self.color_embedding = nn.Parameter(torch.zeros((int(num_views), 12)))


class HostEmbedding(nn.Module):
    def __init__(self, cfg, shapes, dataset=None):
        super().__init__()
        mods = []
        for e in cfg['embedding']['embeddings'].values():
            if e['type'] == 'ray_prediction':
                mods.append(HostRayPrediction(e, shapes))
            elif e['type'] == 'point_prediction':         # cascades: a second MLP, same container shape
                from .scenes import point_mlp_layer_shapes
                mods.append(HostRayPrediction(e, point_mlp_layer_shapes(e)))
            elif e['type'] == 'color_transform':
                mods.append(HostColorTransform((dataset or {}).get('total_images_per_frame', 1)))
            else:
                mods.append(_Empty())
        self.embeddings = nn.ModuleList(mods)


class HostTensorVM(nn.Module):
    """Parameter container shaped like TensorVMSplit (nlf/nets/tensorf_base.py:895-991) or
    TensorVMKeyframeTime (nlf/nets/tensorf_dynamic.py:108-244)."""

    def __init__(self, net_cfg, grid_size, num_keyframes):
        super().__init__()
        self.video = net_cfg['type'] == 'tensor_vm_split_time'
        self.n_den = list(net_cfg.get('n_lamb_sigma', [8, 8, 8]))
        self.n_app = list(net_cfg.get('n_lamb_sh', [24, 24, 24]))
        self.app_dim = int(net_cfg.get('data_dim_color', 27))
        self.num_keyframes = int(num_keyframes)
        self.act = net_cfg.get('fea2denseAct', 'softplus')
        self.alpha_mask_thres = float(net_cfg.get('alpha_mask_thre', 0.001))         # tensorf_base.py:206-208
        self.update_alpha_mask_list = [int(v) for v in net_cfg.get('update_AlphaMask_list', [])]
        self._owner = None                       # weakref to the HipLightfieldModel (set by it): native handle for getDenseAlpha
        # growth schedule of the grids (tensorf_base.py:150-198): log-linear between the start and end resolutions
        self.upsamp_list = [int(v) for v in net_cfg.get('upsamp_list', [])]
        n_up = len(self.upsamp_list) + 1
        steps = lambda a, b: torch.round(torch.exp(torch.linspace(math.log(a), math.log(b), n_up))).long().tolist()[1:]
        self.use_grid_size_upsample = 'grid_size' in net_cfg
        if self.use_grid_size_upsample:
            gs, ge = list(net_cfg['grid_size']['start']), list(net_cfg['grid_size']['end'])
            self.N_voxel_list = [steps(gs[i], ge[i]) for i in range(3)]
        elif 'N_voxel_init' in net_cfg and 'N_voxel_final' in net_cfg:
            self.N_voxel_list = steps(net_cfg['N_voxel_init'], net_cfg['N_voxel_final'])
        else:
            self.N_voxel_list = []
        self.register_buffer('aabb', torch.tensor(to_plain(net_cfg['aabb']), dtype=torch.float32))
        self.register_buffer('gridSize', torch.tensor([int(v) for v in grid_size], dtype=torch.long))
        self.basis_mat = nn.Linear(sum(self.n_app), self.app_dim, bias=False)
        if self.video:
            self.basis_mat_density = nn.Linear(sum(self.n_den), 1, bias=False)
        self.init_svd_volume(grid_size)

    def _dens(self, shape):
        if self.act == 'softplus':
            return 0.1 * torch.randn(shape)
        return 1e-2 * torch.rand(shape).clamp(1e-2, 1e8)

    def init_svd_volume(self, grid_size, device=None):
        """(Re)allocates the planes for `grid_size` with the reference's initialisers."""
        N = [int(v) for v in grid_size]
        device = device if device is not None else self.aabb.device
        self.gridSize = torch.tensor(N, dtype=torch.long, device=device)
        mk = lambda fn, shapes: nn.ParameterList([nn.Parameter(fn(s).to(device)) for s in shapes])
        app = lambda s: 0.1 * torch.randn(s)
        plane = lambda n: [(1, n[i], N[MAT_MODE[i][1]], N[MAT_MODE[i][0]]) for i in range(3)]
        if self.video:
            time = lambda n: [(1, n[i], self.num_keyframes, N[MAT_MODE_TIME[i][0]]) for i in range(3)]
            self.density_plane_space = mk(self._dens, plane(self.n_den))
            self.density_plane_time = mk(self._dens, time(self.n_den))
            self.app_plane_space = mk(app, plane(self.n_app))
            self.app_plane_time = mk(app, time(self.n_app))
        else:
            line = lambda n: [(1, n[i], N[VEC_MODE[i]], 1) for i in range(3)]
            self.density_plane = mk(self._dens, plane(self.n_den))
            self.density_line = mk(self._dens, line(self.n_den))
            self.app_plane = mk(app, plane(self.n_app))
            self.app_line = mk(app, line(self.n_app))

    def set_iter(self, i):
        """TensorBase.set_iter (tensorf_base.py:510-552): in training mode, at the iterations of update_AlphaMask_list the
        occupancy mask is rebuilt (at most 200^3) and, the first time, the grids are shrunk to it; at the iterations of
        upsamp_list the grids grow to the next resolution of N_voxel_list / grid_size."""
        self.cur_iter = i
        if not self.training:
            return
        if i in self.update_alpha_mask_list:
            reso = tuple(int(v) for v in self.gridSize.tolist())
            if reso[0] > 200:
                reso = (200, 200, 200)
            new_aabb = self.updateAlphaMask(reso)
            if i == self.update_alpha_mask_list[0]:
                self.shrink(new_aabb)
        if i in self.upsamp_list and self.N_voxel_list:
            if self.use_grid_size_upsample:
                reso = [self.N_voxel_list[a].pop(0) for a in range(3)]
            else:
                reso = n_to_reso(self.N_voxel_list.pop(0), self.aabb.detach().cpu().tolist())
            self.upsample_volume_grid(reso)

    # -- occupancy / grid management of the training loop (TensorBase.set_iter, nlf/nets/tensorf_base.py:510-530) -------
    @torch.no_grad()
    def getDenseAlpha(self, grid_size):
        """TensorBase.getDenseAlpha (tensorf_base.py:381-401) / TensorVMKeyframeTime.getDenseAlpha (tensorf_dynamic.py:499-536):
        alpha = 1 - exp(-sigma * 0.01) on a dense (n0, n1, n2) lattice of the box (keyframe nets: the maximum over the
        frames), points the current mask rejects get 0.  One HIP launch (hr_dense_alpha) on the owner's native model."""
        import ctypes as C
        owner = self._owner() if getattr(self, '_owner', None) is not None else None
        if owner is None:
            raise RuntimeError('getDenseAlpha needs the owning HipLightfieldModel (its native handle)')
        h = owner.native()
        n = [int(v) for v in grid_size]
        dev = self.aabb.device
        out = torch.empty(n, dtype=torch.float32, device=dev)
        L = _lib.load()
        vol = getattr(self, 'alpha_volume', None)
        box = self.alpha_aabb.detach().cpu().reshape(-1).tolist() if vol is not None else [0.0] * 6
        pn = [vol.shape[2], vol.shape[1], vol.shape[0]] if vol is not None else [0, 0, 0]
        with torch.cuda.device(dev):
            _lib.check(L.hr_dense_alpha(h, (C.c_int32 * 3)(*n), C.c_float(0.01), int(owner.dataset.get('num_frames', 1)),
                                        C.c_void_p(vol.contiguous().data_ptr()) if vol is not None else C.c_void_p(0),
                                        (C.c_int32 * 3)(*pn), (C.c_float * 6)(*box), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'hr_dense_alpha')
        return out

    @torch.no_grad()
    def updateAlphaMask(self, grid_size=(200, 200, 200)):
        """TensorBase.updateAlphaMask (tensorf_base.py:403-429): dense alpha -> 3x3x3 max-pool -> threshold -> the mask
        volume (stored like AlphaGridMask does, utils/tensorf_utils.py:459-475) and the bounding box of what is left."""
        import torch.nn.functional as F
        n = [int(v) for v in grid_size]
        alpha = self.getDenseAlpha(n)
        dev = alpha.device
        samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, n[0]), torch.linspace(0, 1, n[1]), torch.linspace(0, 1, n[2]),
                                             indexing='ij'), -1).to(dev)
        dense_xyz = self.aabb[0] * (1 - samples) + self.aabb[1] * samples
        dense_xyz = dense_xyz.transpose(0, 2).contiguous()
        alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
        alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(n[::-1])
        thres = float(self.alpha_mask_thres)
        alpha = (alpha >= thres).to(torch.float32)
        self.register_buffer('alpha_volume', alpha, persistent=False)            # (n2, n1, n0)
        self.register_buffer('alpha_aabb', self.aabb.clone(), persistent=False)
        valid = dense_xyz[alpha > 0.5]
        return torch.stack((valid.amin(0), valid.amax(0)))

    @torch.no_grad()
    def shrink(self, new_aabb):
        """TensorVMSplit.shrink (tensorf_base.py:1191-1232) / TensorVMKeyframeTime.shrink (tensorf_dynamic.py:444-497):
        crops every plane and line to the texels that cover `new_aabb` and moves the box to those texels."""
        new_aabb = new_aabb.to(self.aabb)
        grid = self.gridSize.to(self.aabb.device)
        units = (self.aabb[1] - self.aabb[0]) / (grid - 1)
        t_l, b_r = (new_aabb[0] - self.aabb[0]) / units, (new_aabb[1] - self.aabb[0]) / units
        t_l, b_r = torch.round(torch.round(t_l)).long(), torch.round(b_r).long() + 1
        b_r = torch.stack([b_r, grid]).amin(0)
        tl, br = t_l.tolist(), b_r.tolist()
        P = lambda t: nn.Parameter(t.contiguous())
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            if self.video:
                t0 = MAT_MODE_TIME[i][0]
                for space, time in ((self.density_plane_space, self.density_plane_time), (self.app_plane_space, self.app_plane_time)):
                    space[i] = P(space[i].data[..., tl[m1]:br[m1], tl[m0]:br[m0]])
                    time[i] = P(time[i].data[..., :, tl[t0]:br[t0]])
            else:
                v = VEC_MODE[i]
                for plane, line in ((self.density_plane, self.density_line), (self.app_plane, self.app_line)):
                    line[i] = P(line[i].data[..., tl[v]:br[v], :])
                    plane[i] = P(plane[i].data[..., tl[m1]:br[m1], tl[m0]:br[m0]])
        mask_grid = torch.tensor([self.alpha_volume.shape[2], self.alpha_volume.shape[1], self.alpha_volume.shape[0]], device=grid.device)
        if not torch.all(mask_grid == grid):
            t_l_r, b_r_r = t_l / (grid - 1), (b_r - 1) / (grid - 1)
            correct = torch.zeros_like(new_aabb)
            correct[0] = (1 - t_l_r) * self.aabb[0] + t_l_r * self.aabb[1]
            correct[1] = (1 - b_r_r) * self.aabb[0] + b_r_r * self.aabb[1]
            new_aabb = correct
        self.aabb = new_aabb.clone()                 # a fresh buffer: the owner sees a new box and rebuilds its native model
        self.gridSize = (b_r - t_l).to(self.gridSize)

    # -- regularizers, as the reference's TensoRF regularizer calls them (nlf/regularizers/tensorf.py:57-92) ----------
    def _reg_planes(self):
        if self.video:
            return self.density_plane_space, self.density_plane_time, self.app_plane_space
        return self.density_plane, self.density_line, self.app_plane

    def density_L1(self):
        """tensorf_base.py:1024-1035 / tensorf_dynamic.py:246-259."""
        from .train import l1_mean
        da, db, _ = self._reg_planes()
        total = 0
        for i in range(3):
            if da[i].shape[1] == 0:
                continue
            total = total + l1_mean(da[i]) + l1_mean(db[i])
        return total

    def TV_loss_density(self, reg):
        """tensorf_base.py:1037-1046 / tensorf_dynamic.py:261-272; `reg` is the reference's TVLoss module (only its weight
        is read: the differences are computed by hr_plane_reg_forward)."""
        from .train import tv_loss
        da, db, _ = self._reg_planes()
        total = 0
        for i in range(3):
            if da[i].shape[1] == 0 or (self.video and db[i].shape[1] == 0):
                continue
            total = total + tv_loss(da[i], getattr(reg, 'TVLoss_weight', 1)) * 1e-2
        return total

    def TV_loss_app(self, reg):
        """tensorf_base.py:1048-1057 / tensorf_dynamic.py:274-285."""
        from .train import tv_loss
        da, db, aa = self._reg_planes()
        total = 0
        for i in range(3):
            if (aa[i].shape[1] == 0) if not self.video else (da[i].shape[1] == 0 or db[i].shape[1] == 0):
                continue
            total = total + tv_loss(aa[i], getattr(reg, 'TVLoss_weight', 1)) * 1e-2
        return total

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """TensorVMSplit.upsample_volume_grid (nlf/nets/tensorf_base.py:1152-1188) /
        TensorVMKeyframeTime.upsample_volume_grid (nlf/nets/tensorf_dynamic.py:395-441): every plane and line is
        resized bilinearly (align_corners=True) to the new grid by hr_upsample_plane; planes of a video net whose density
        plane has no components are re-created as zeros, as in the reference.  Parameters must live on the HIP device."""
        import ctypes as C
        L = _lib.load()
        N = [int(v) for v in res_target]

        def resize(p, h2, w2):
            src = p.data.contiguous().float()
            if src.device.type != 'cuda':
                raise RuntimeError('upsample_volume_grid runs on the HIP device; there is no CPU path')
            _, c, h, w = src.shape
            dst = torch.empty((1, c, h2, w2), dtype=torch.float32, device=src.device)
            with torch.cuda.device(src.device):
                _lib.check(L.hr_upsample_plane(C.c_void_p(src.data_ptr()), c, h, w, C.c_void_p(dst.data_ptr()), h2, w2,
                                               C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)), 'hr_upsample_plane')
            return nn.Parameter(dst)

        for i in range(3):
            m0, m1 = MAT_MODE[i]
            if self.video:
                t0 = MAT_MODE_TIME[i][0]
                empty = self.density_plane_space[i].shape[1] == 0
                for space, time in ((self.app_plane_space, self.app_plane_time), (self.density_plane_space, self.density_plane_time)):
                    if empty:
                        space[i] = nn.Parameter(space[i].data.new_zeros(1, space[i].shape[1], N[m1], N[m0]))
                        time[i] = nn.Parameter(time[i].data.new_zeros(1, time[i].shape[1], self.num_keyframes, N[t0]))
                    else:
                        space[i] = resize(space[i], N[m1], N[m0])
                        time[i] = resize(time[i], self.num_keyframes, N[t0])
            else:
                for plane, line in ((self.app_plane, self.app_line), (self.density_plane, self.density_line)):
                    if plane[i].shape[1] > 0:
                        plane[i] = resize(plane[i], N[m1], N[m0])
                    if line[i].shape[1] > 0:
                        line[i] = resize(line[i], N[VEC_MODE[i]], 1)
        self.gridSize = torch.tensor(N, dtype=torch.long, device=self.gridSize.device)


class HostColorModel(nn.Module):
    def __init__(self, net_cfg, grid_size, num_keyframes):
        super().__init__()
        self.net = HostTensorVM(net_cfg, grid_size, num_keyframes)

    def set_iter(self, i):
        self.net.set_iter(i)


class HipLightfieldModel(nn.Module):
    """Drop-in for model_dict['lightfield'] (nlf/models/models.py:141-143)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        from .scenes import mlp_layer_shapes
        self.cfg = cfg
        system = kwargs.get('system')
        self.dataset = kwargs['dataset'] if 'dataset' in kwargs else dataset_scalars_from_system(system)
        # arithmetic of the MLP GEMMs: 'auto' (= 'f16x3' where its fp16 range is proven at finalize, else 'bf16x3') | 'f16x3' | 'bf16x3' | 'f16f8' | 'f16x2' | 'fp32' (plan.compile_config)
        self.mlp_precision = kwargs.get('mlp_precision', 'auto')
        self.grid_dtype = kwargs.get('grid_dtype', 'fp32')     # 'fp16': half-precision texels (viewer path)
        # execution plan of render() (hr_model_set_option): frame kernel on/off, its sample wavefronts (None: library default)
        # opt-in occupancy early-reject (hr_model_set_occupancy): samples the colour net's AlphaGridMask rejects are neither
        # gathered nor composited -- the reference's own test at tensorf_no_sample.py:171-177, which it ships disabled
        self.use_occupancy = bool(kwargs.get('use_occupancy', False))
        self._occ_key = None
        self.frame_kernel = self._frame_mode(kwargs.get('frame_kernel', False))
        self.sample_waves = kwargs.get('sample_waves')
        self.train_deterministic = bool(kwargs.get('train_deterministic', False))
        self.train_fused_mlp = bool(kwargs.get('train_fused_mlp', False))
        net = cfg['color']['net']
        if 'grid_size' in kwargs and kwargs['grid_size'] is not None:
            grid = list(kwargs['grid_size'])
        elif 'grid_size' in net:                                   # tensorf_base.py:150-154
            grid = list(net['grid_size']['start'])
        else:
            grid = n_to_reso(net['N_voxel_init'], to_plain(net['aabb']))
        self.param = _Dummy()
        self.embedding_model = HostEmbedding(cfg, mlp_layer_shapes(cfg), self.dataset)
        self.color_model = HostColorModel(net, grid, self.dataset['num_keyframes'])
        import weakref
        self.color_model.net._owner = weakref.ref(self)
        self.cur_iter = None           # training iteration of the activation / PE schedules; None = converged (see set_iter)
        self._native = None
        self._native_key = None
        self._native_cfg = None        # bytes of the hr_config the native handle currently holds
        self._sched_built = None       # cur_iter that configuration was compiled at
        self._native_box = None        # (data_ptr, version) of the net's aabb buffer the handle was created for
        self._coarse_hc = None         # hr_config of a cascade's coarse level (None for single-level models)
        # fail on configurations outside the supported path now, not at the first render
        self._compile(grid)

    def upsample_volume_grid(self, res_target):
        """Grows the feature grids like the reference's training loop does (nlf/__init__.py upsampling hooks ->
        color_model.net.upsample_volume_grid); the native model is rebuilt for the new size at the next render."""
        self.color_model.net.upsample_volume_grid(res_target)
        self._native_key = None

    def _compile(self, grid):
        """-> (coarse hr_config or None, hr_config of the level that renders), schedules evaluated at cur_iter."""
        # the box is the net's buffer, not the YAML's value: `shrink` replaces it during training and checkpoints carry it
        box = self.color_model.net.aabb
        key = (box.data_ptr(), box._version)
        if getattr(self, '_box_host', (None, None))[0] != key:        # one device->host copy per change of the box, not per call
            self._box_host = (key, box.detach().cpu().numpy())
        aabb = self._box_host[1]
        return compile_model(self.cfg, self.dataset, grid, self.mlp_precision, self.grid_dtype, iteration=self.cur_iter, aabb=aabb)

    # -- reference surface ---------------------------------------------------------
    def set_iter(self, i):
        """LightfieldModel.set_iter (what INRSystem.set_train_iter calls every step, nlf/__init__.py:608-614): the EaseValue
        activations (nlf/activations.py:462-496) and WindowedPE weights (nlf/pe.py:166-208) of the model are evaluated at
        training iteration `i`.  Render / test call it with 1e7 (nlf/__init__.py:582-583), where every weight is 1; a model
        that was never told an iteration is in that converged state too.  Only constants of the compiled configuration
        change: the next call swaps them in with hr_model_update_config, no weights are re-packed."""
        self.cur_iter = i
        self.color_model.set_iter(i)

    @property
    def grid_size(self):
        # gridSize is a buffer and lives on the device with the model: read it back only when it changed (forward_train asks every step --
        # a device-to-host copy there stalls the host behind the queue and cannot be captured into a graph)
        # The cache holds the TENSOR it read (shrink / upsample_volume_grid / load_state_dict replace gridSize by fresh tensors whose _version is 0
        # again and whose storage the allocator may hand out at a freed tensor's address: identity, not data_ptr, is the key -- ADVICE r4)
        gs = self.color_model.net.gridSize
        cached = getattr(self, '_grid_size_host', None)
        if cached is None or cached[0] is not gs or cached[1] != gs._version:
            cached = self._grid_size_host = (gs, gs._version, [int(v) for v in gs.tolist()])
        return list(cached[2])

    def load_state_dict(self, state_dict, strict=True):
        """Accepts reference checkpoints: strips a leading `render_fn.model.` / `model.`,
        re-allocates the planes when the checkpoint's gridSize differs and reshapes plane
        tensors (nlf/__init__.py:433-479)."""
        sd = {}
        for k, v in state_dict.items():
            for pre in ('render_fn.model.', 'model.'):
                if k.startswith(pre):
                    k = k[len(pre):]
                    break
            sd[k] = v
        # the alpha mask a trained checkpoint carries (nlf/__init__.py:437-447,472-479) is kept but not read: the colour
        # net's forward never consults it (`if self.alphaMask is not None and False`, tensorf_no_sample.py:171)
        mask = {k: sd.pop(k) for k in list(sd) if 'alpha_aabb' in k or 'alpha_volume' in k}
        if mask:
            net = self.color_model.net
            net.alpha_mask_state = mask
            # the occupancy early-reject (use_occupancy) reads net.alpha_volume / net.alpha_aabb: materialise them from the checkpoint's
            # AlphaGridMask (alphaMask.alpha_volume (1, 1, D, H, W) or flat, alphaMask.alpha_aabb; tensorf_base.py:1139-1168)
            vol = next((v for k, v in mask.items() if k.endswith('alpha_volume')), None)
            box = next((v for k, v in mask.items() if k.endswith('alpha_aabb')), None)
            if vol is not None and box is not None:
                vol = vol.detach().float()
                if vol.dim() == 5:
                    vol = vol[0, 0]
                if vol.dim() == 3:
                    dev = net.aabb.device
                    net.register_buffer('alpha_volume', vol.to(dev).contiguous(), persistent=False)
                    net.register_buffer('alpha_aabb', box.detach().float().reshape(2, 3).to(dev), persistent=False)
        gs = sd.get('color_model.net.gridSize')
        if gs is not None and [int(x) for x in gs.tolist()] != self.grid_size:
            self.color_model.net.init_svd_volume([int(x) for x in gs.tolist()])
        own = self.state_dict()
        for k in list(sd.keys()):
            if k in own and torch.is_tensor(sd[k]) and sd[k].shape != own[k].shape and sd[k].numel() == own[k].numel():
                sd[k] = sd[k].view(own[k].shape)
        self._native_key = None
        return super().load_state_dict(sd, strict=strict)

    # -- native side -------------------------------------------------------------------
    def _tensors(self):
        own = dict(self.named_parameters())
        coarse, hc = self._compile(self.grid_size)
        types = [e['type'] for e in self.cfg['embedding']['embeddings'].values()]
        idx = {'idx': types.index('ray_prediction'),
               'pp_idx': types.index('point_prediction') if 'point_prediction' in types else -1,
               'ct_idx': types.index('color_transform') if 'color_transform' in types else -1}
        return coarse, hc, [(abi, own[key.format(**idx)]) for abi, key in upload_names(hc, coarse)]

    def _param_key(self):
        # cheap fingerprint of everything the native model was built from: in-place
        # updates bump ._version, re-allocations change data_ptr/shape
        net = self.color_model.net
        return (self.mlp_precision, self.grid_dtype, net.aabb.data_ptr(), net.aabb._version) + \
            tuple((p.data_ptr(), p._version, tuple(p.shape)) for p in self.parameters())

    def _sync_schedule(self, hc=None, coarse=None):
        """Hands the configuration compiled at cur_iter to an existing native handle if it differs from what it holds."""
        import ctypes as C
        if hc is None:
            coarse, hc = self._compile(self.grid_size)
        blob = bytes(hc) + (bytes(coarse) if coarse is not None else b'')
        if blob != self._native_cfg:
            if coarse is not None:
                # hr_model_update_config does not take cascades (two nested configurations): re-create the handle with
                # the constants of this iteration and upload again
                _lib.load().hr_model_destroy(self._native)
                self._native = None
                self._render_calls = 0
                self._native_key = None
                self.native()
                return
            dev = next(self.parameters()).device
            with torch.cuda.device(dev):
                _lib.check(_lib.load().hr_model_update_config(self._native, C.byref(hc), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                           'hr_model_update_config')
            self._native_cfg = blob
        self._hc = hc
        self._sched_built = self.cur_iter

    def native(self):
        """Returns the hr_model handle, (re)uploading weights if any parameter changed."""
        key = self._param_key()
        if self._native is not None and key == self._native_key:
            if self._sched_built != self.cur_iter:
                self._sync_schedule()
            return self._native
        coarse, hc, tensors = self._tensors()
        L = _lib.load()
        dev = tensors[0][1].device
        if dev.type != 'cuda':
            raise RuntimeError('HipLightfieldModel must live on a HIP device (model.cuda()); there is no CPU path')
        import ctypes as C
        def create():
            h = C.c_void_p()
            if coarse is None:
                _lib.check(L.hr_model_create(C.byref(hc), C.byref(h)), 'hr_model_create')
            else:
                _lib.check(L.hr_model_create_cascade(C.byref(coarse), C.byref(hc), C.byref(h)), 'hr_model_create_cascade')
            self._native = h
            self._occ_key = None                   # a fresh handle holds no occupancy volume
            self._apply_options()
            if getattr(self, '_reserve', None):
                _lib.check(L.hr_model_reserve(h, self._reserve), 'hr_model_reserve')
            self._native_grid = self.grid_size
            self._native_box = (self.color_model.net.aabb.data_ptr(), self.color_model.net.aabb._version)
            self._native_cfg = bytes(hc) + (bytes(coarse) if coarse is not None else b'')

        with torch.cuda.device(dev):
            if self._native is None:
                create()
            elif self._native_grid != self.grid_size or self._hc.mlp_precision != hc.mlp_precision \
                    or self._hc.grid_dtype != hc.grid_dtype or list(self._hc.aabb) != list(hc.aabb):
                L.hr_model_destroy(self._native)
                create()
            torch.cuda.current_stream().synchronize()
            for abi, t in tensors:
                t = t.detach().contiguous().float()
                _lib.check(L.hr_model_upload(self._native, abi.encode(), C.c_void_p(t.data_ptr()),
                                             t.numel() * 4), f'hr_model_upload({abi})')
            _lib.check(L.hr_model_finalize(self._native), 'hr_model_finalize')
        self._native_key = key
        self._hc = hc
        self._coarse_hc = coarse
        self._sync_schedule(hc, coarse)          # an existing handle may still hold another iteration's constants
        return self._native

    def _sync_occupancy(self):
        """Hands the colour net's current mask volume to the native model (or clears it)."""
        import ctypes as C
        net = self.color_model.net
        vol = getattr(net, 'alpha_volume', None) if self.use_occupancy else None
        key = None if vol is None else (vol.data_ptr(), vol._version, tuple(vol.shape), net.alpha_aabb.data_ptr(), net.alpha_aabb._version)
        if key == self._occ_key or self._native is None:
            return
        L = _lib.load()
        dev = next(self.parameters()).device
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if vol is None:
                _lib.check(L.hr_model_set_occupancy(self._native, C.c_void_p(0), None, None, stream), 'hr_model_set_occupancy')
            else:
                v = vol.detach().contiguous().float().to(dev)
                n = (C.c_int32 * 3)(v.shape[2], v.shape[1], v.shape[0])
                box = (C.c_float * 6)(*net.alpha_aabb.detach().cpu().reshape(-1).tolist())
                _lib.check(L.hr_model_set_occupancy(self._native, C.c_void_p(v.data_ptr()), n, box, stream), 'hr_model_set_occupancy')
        self._occ_key = key

    def set_occupancy(self, enable=True):
        """Turns the occupancy early-reject of render() on or off (needs a mask: updateAlphaMask or a checkpoint's)."""
        if enable and getattr(self.color_model.net, 'alpha_volume', None) is None:
            raise RuntimeError('set_occupancy(True): the colour net has no alpha mask volume (run updateAlphaMask, or load a checkpoint that carries one)')
        self.use_occupancy = bool(enable)
        if self._native is not None:
            self._sync_occupancy()

    def _apply_options(self):
        L = _lib.load()
        _lib.check(L.hr_model_set_option(self._native, _lib.HR_OPT_FRAME_KERNEL, int(self.frame_kernel)), 'hr_model_set_option')
        if self.sample_waves is not None:
            _lib.check(L.hr_model_set_option(self._native, _lib.HR_OPT_SAMPLE_WAVES, int(self.sample_waves)), 'hr_model_set_option')
        _lib.check(L.hr_model_set_option(self._native, _lib.HR_OPT_TRAIN_DETERMINISTIC, int(self.train_deterministic)), 'hr_model_set_option')

    @staticmethod
    def _frame_mode(v):
        """HR_OPT_FRAME_KERNEL: False / 0 = two kernels (default), True / 1 = the frame kernel for the static nets it fits on 64-ray tiles,
        2 = wherever it fits"""
        return 2 if (v == 2 and v is not True) else int(bool(v))

    def set_train_deterministic(self, enable=True):
        """HR_OPT_TRAIN_DETERMINISTIC: the training step's gradient sums as 64-bit fixed point through integer atomics -- two runs of the
        same step agree bit for bit (the default, fp32 atomics, is faster and reproducible to rounding only)."""
        self.train_deterministic = bool(enable)
        if self._native is not None:
            self._apply_options()

    def set_execution(self, frame_kernel=None, sample_waves=None):
        """Chooses how render() is laid out on the device (images are bit-identical under every setting):
        frame_kernel False = the two-kernel path through the HBM workspace (default), True = the persistent frame kernel for the static
        nets (64-ray tiles), 2 = wherever the model fits it; sample_waves 4 | 8.
        The frame kernel is a one-pass plan: a model on the verified fast path (mlp_precision 'auto' / 'f16f8v', two passes over the HBM
        workspace) keeps the two-kernel path whatever is asked for here -- choose mlp_precision 'f16x3' (or plain 'f16f8') with it."""
        if frame_kernel is not None:
            self.frame_kernel = self._frame_mode(frame_kernel)
        if sample_waves is not None:
            self.sample_waves = int(sample_waves)
        if self._native is not None:
            self._apply_options()
            if self.frame_kernel and self.mlp_verified():
                import warnings
                warnings.warn("hyperreel_amd: frame_kernel was requested for a model on the verified fast path (mlp_precision 'auto' resolved to "
                              "f16f8 + verification), which is a two-pass plan: the request has no effect; use mlp_precision='f16x3' with it")

    def frame_kernel_active(self):
        """True when render() of this model runs as the single persistent frame kernel (head tile in LDS)."""
        import ctypes as C
        v = C.c_int32(0)
        _lib.check(_lib.load().hr_model_get_option(self.native(), _lib.HR_OPT_FRAME_KERNEL_ACTIVE, C.byref(v)), 'hr_model_get_option')
        return bool(v.value)

    def chunk_rays(self):
        """Rays per launch of the head workspace (hr_model_reserve / the finalize default): the most hr_stage_* accept."""
        return self._get_option(_lib.HR_OPT_CHUNK_RAYS)

    def _get_option(self, opt):
        import ctypes as C
        v = C.c_int32(0)
        _lib.check(_lib.load().hr_model_get_option(self.native(), opt, C.byref(v)), 'hr_model_get_option')
        return int(v.value)

    def mlp_precision_active(self):
        """The arithmetic the MLP kernels run ('auto' resolved by the library's activation-range calibration)."""
        return {0: 'fp32', 1: 'bf16x3', 2: 'f16x3', 3: 'f16x2', 5: 'f16f8'}[self._get_option(_lib.HR_OPT_MLP_PRECISION_ACTIVE)]

    def mlp_verified(self):
        """True when render() runs the verified fast path: f16f8 first, then the rays with a comparison at risk again with the f16x3 tiles
        (HR_MLP_F16F8V; what 'auto' resolves to for a plain ray MLP with at most 64 samples per ray)."""
        return bool(self._get_option(_lib.HR_OPT_MLP_VERIFIED))

    def redo_count(self):
        """Rays the last render() listed for its second pass (synchronises)."""
        return self._get_option(_lib.HR_OPT_REDO_COUNT)

    def wide_count(self):
        """Rays the last render() passed on to its third pass (bf16x3 tiles) because an activation left the IEEE-half range (synchronises)."""
        return self._get_option(_lib.HR_OPT_WIDE_COUNT)

    def redo_overflowed(self):
        """Sticky: a render() listed more rays than a call's list holds (max(32768, B / 16)); the excess kept their first-pass pixels
        (_overflow_guard then re-decides the arithmetic on those rays and renders the batch again)."""
        return bool(self._get_option(_lib.HR_OPT_REDO_OVERFLOW))

    def verify_info(self):
        """hr_model_verify_info as a dict: the band of the verified fast path for THIS model and the f16f8-vs-f16x3 measurement on the
        calibration rays it was derived from (band = max(floor, 4 x the largest difference of a distance / length / point))."""
        import ctypes as C
        v = _lib.hr_verify_info()
        _lib.check(_lib.load().hr_model_verify_info(self.native(), C.byref(v)), 'hr_model_verify_info')
        return {k: getattr(v, k) for k, _ in v._fields_}

    def mlp_overflowed(self):
        """True when an fp16-split kernel saw an activation at the IEEE-half range on a rendered ray (sticky; synchronises)."""
        return bool(self._get_option(_lib.HR_OPT_MLP_OVERFLOW))

    def mlp_f8_saturated(self):
        """mlp_precision 'f16f8': True when a hidden activation of a rendered ray was beyond the range of its fp8 image (sticky; synchronises).
        Nothing overflowed -- that ray's correction products were computed from saturated images; `calibrate(rays)` moves the exponents."""
        return bool(self._get_option(_lib.HR_OPT_MLP_F8_SATURATED))

    def calibrate(self, rays):
        """Re-decides the MLP arithmetic on the caller's rays (hr_model_calibrate); returns the per-layer activation maxima."""
        import ctypes as C
        self.native()
        rays = self._check_rays(rays)
        out = (C.c_float * 8)()
        with torch.cuda.device(rays.device):
            stream = C.c_void_p(torch.cuda.current_stream(rays.device).cuda_stream)
            _lib.check(_lib.load().hr_model_calibrate(self.native(), C.c_void_p(rays.data_ptr()), rays.shape[0], out, stream), 'hr_model_calibrate')
        return [float(v) for v in out[:int(self._hc.mlp_layers)]]

    def reserve(self, rays_per_chunk):
        """Sizes the native workspace; remembered, so that a handle that is re-created (grid growth, cascade schedules) gets it again."""
        self._reserve = int(rays_per_chunk)
        L = _lib.load()
        _lib.check(L.hr_model_reserve(self.native(), self._reserve), 'hr_model_reserve')

    def device_bytes(self):
        return int(_lib.load().hr_model_device_bytes(self.native()))

    def __del__(self):
        try:
            if self._native is not None:
                _lib.load().hr_model_destroy(self._native)
                self._native = None
                self._render_calls = 0
        except Exception:
            pass

    # -- rendering ---------------------------------------------------------------------
    def _check_rays(self, rays):
        hc = self._hc
        if rays.device.type != 'cuda':
            raise RuntimeError('rays must be on the HIP device; there is no CPU path')
        if rays.dim() != 2 or rays.shape[1] < hc.ray_dim:
            raise ValueError(f'rays must be (B,{hc.ray_dim}), got {tuple(rays.shape)}')
        # the C ABI takes no row stride: extra columns are cut (6-column nets ignore camera id / time, rendering.py)
        return rays[:, :hc.ray_dim].contiguous().float()

    def render(self, rays, want=(), out=None, frame_time=None):
        """rays (B, 6|8) on the HIP device -> dict with 'rgb' (B,3) and any of
        'distances' (B,Z), 'points' (B,Z,3), 'sigma' (B,Z), 'render_weights' (B,Z),
        'head' (B,Z*P) listed in `want`.  out: an existing (B,3) float32 device tensor to render into.
        frame_time: the caller's statement that every ray carries this time (one frame of a keyframe net): hr_render_frame, which reads
        one row of each time plane instead of blending two (images agree with the general path to ~1e-6, not bit for bit)."""
        import ctypes as C
        h = self.native()
        self._sync_occupancy()
        L = _lib.load()
        rays = self._check_rays(rays)
        B = rays.shape[0]
        hc = self._hc
        Z = hc.z_channels
        if out is not None and (out.shape != (B, 3) or out.dtype != torch.float32 or out.device != rays.device or not out.is_contiguous()):
            raise ValueError('out must be a contiguous (B, 3) float32 tensor on the rays\' device')
        out = {'rgb': out if out is not None else torch.empty((B, 3), dtype=torch.float32, device=rays.device)}
        stream = C.c_void_p(torch.cuda.current_stream(rays.device).cuda_stream)
        with torch.cuda.device(rays.device):
            if not want:
                for attempt in (0, 1):
                    if frame_time is not None:
                        _lib.check(L.hr_render_frame(h, C.c_void_p(rays.data_ptr()), B, float(frame_time), C.c_void_p(out['rgb'].data_ptr()), stream),
                                   'hr_render_frame')
                    else:
                        _lib.check(L.hr_render(h, C.c_void_p(rays.data_ptr()), B, C.c_void_p(out['rgb'].data_ptr()), stream), 'hr_render')
                    if attempt == 1 or not self._overflow_guard(rays):
                        break
                return out
            f = hr_fields()
            shapes = {'distances': (B, Z), 'points': (B, Z, 3), 'sigma': (B, Z), 'render_weights': (B, Z),
                      'head': (B, Z * hc.preds_per_z)}
            slots = {'distances': 'distances_dev', 'points': 'points_dev', 'sigma': 'sigma_dev',
                     'render_weights': 'weights_dev', 'head': 'head_dev'}
            for k in want:
                out[k] = torch.zeros(shapes[k], dtype=torch.float32, device=rays.device)
                setattr(f, slots[k], out[k].data_ptr())
            _lib.check(L.hr_render_fields(h, C.c_void_p(rays.data_ptr()), B, C.c_void_p(out['rgb'].data_ptr()),
                                          C.byref(f), stream), 'hr_render_fields')
        return out

    def _overflow_guard(self, rays):
        """The fp16 split arithmetic ('auto' -> f16x3) was chosen on CALIBRATION rays (hr_model_finalize: synthetic origins in and around
        the scene box); real cameras may stand further out, and activations beyond the IEEE-half range would render inf / NaN where the
        reference's fp32 BaseMLP (nlf/nets/mlp.py:159-172) does not.  The kernels raise a sticky bit when that happens; it is read here
        on the model's first 16 render calls and every 256th after (one 4-byte read behind a synchronise; never inside a stream
        capture).  If set: the arithmetic is re-decided on the offending rays -- 'auto' re-packs as bf16x3 (fp32 exponent range), a forced
        fp16 mode raises HipRangeError -- and the caller renders the batch again.  Returns True when it did so."""
        n = self._render_calls = getattr(self, '_render_calls', 0) + 1
        if not (n <= 16 or n % 256 == 0) or torch.cuda.is_current_stream_capturing():
            return False
        active = self._get_option(_lib.HR_OPT_MLP_PRECISION_ACTIVE)
        if active not in (2, 3, 5):                                      # f16x3, f16x2, f16f8
            return False
        import warnings
        if self.mlp_overflowed():
            warnings.warn('hyperreel_amd: an MLP activation reached the IEEE-half range on rendered rays (the fp16 split arithmetic had been chosen '
                          'on calibration rays); re-deciding the arithmetic on these rays and rendering the batch again')
            try:
                self.calibrate(rays)
            except RuntimeError as e:
                # a point_prediction cascade cannot be calibrated on the caller's rays (the point MLP's rows are internal: hr_model_calibrate
                # refuses): 'auto' re-creates the native model with the fp32-range split, a forced fp16 mode is refused by name
                if 'cascades' not in str(e):
                    raise
                if self.mlp_precision != 'auto':
                    raise _lib.HipRangeError(f'mlp_precision {self.mlp_precision!r} overflowed the IEEE-half range on rendered rays of a point_prediction '
                                             'cascade; use mlp_precision="auto" or "bf16x3"') from e
                self.mlp_precision = 'bf16x3'
                self._native_key = None
                self.native()
            return True
        if active == 5 and self.mlp_verified() and self.redo_overflowed():
            # verified fast path: more than a sixteenth of the batch had a comparison inside its margin -- rays unlike the calibration's (which
            # gives the fast path up above a twentieth).  The excess kept their unverified pixels: measure the band on THESE rays (the library
            # falls back to f16x3 when they list too many) and render again; a model that still overflows leaves the fast path for good
            warnings.warn('hyperreel_amd: the verified fast path listed more rays than a call holds; re-calibrating on these rays and rendering the batch again')
            self.calibrate(rays)
            if self.mlp_verified() and self.verify_info()['listed_frac'] > 0.05:
                self.mlp_precision = 'f16x3'
                self._native_key = None
                self.native()
            return True
        if active == 5 and self.mlp_f8_saturated():
            # f16f8: the fp8 images of a layer's output are scaled from the calibration's largest activation of that layer.  Beyond 16x that they
            # saturate (finite, less accurate).  Where the model can be calibrated on the caller's rays, do so and render again; a cascade's
            # point MLP sees internal rows (hr_model_calibrate refuses it): say so once and carry on
            if getattr(self, '_f8_cannot_calibrate', False):
                return False
            try:
                self.calibrate(rays)
            except RuntimeError as e:
                if 'cascades' not in str(e):
                    raise
                self._f8_cannot_calibrate = True
                warnings.warn('hyperreel_amd: mlp_precision f16f8 saturated the fp8 image of an activation on rendered rays and this model (a cascade) '
                              'cannot be re-calibrated on them: those rays carry less accurate correction products (mlp_precision f16x3 has none)')
                return False
            warnings.warn('hyperreel_amd: mlp_precision f16f8 saturated the fp8 image of an activation on rendered rays; moved the exponents to these '
                          'rays and rendering the batch again')
            return True
        return False

    def generate_rays(self, pose, K, width, height, time=None, cam_id=0.0, pixel_range=None, device=None):
        """get_coords_from_camera (datasets/base.py:485-518) on the device: 3x4 camera-to-world
        `pose`, 3x3 intrinsics `K` -> rays (n, 6|8) for pixels [lo, hi) of the row-major image
        (whole image by default).  80 bytes cross the PCIe bus instead of the ray list."""
        import ctypes as C
        import numpy as np
        from .plan import hr_camera
        self.native()
        L = _lib.load()
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        lo, hi = (0, int(width) * int(height)) if pixel_range is None else (int(pixel_range[0]), int(pixel_range[1]))
        cam = hr_camera()
        p = np.asarray(pose, np.float32)[:3, :4].reshape(-1)
        for i in range(12):
            cam.c2w[i] = float(p[i])
        Km = np.asarray(K, np.float32)
        cam.fx, cam.fy, cam.cx, cam.cy = float(Km[0, 0]), float(Km[1, 1]), float(Km[0, 2]), float(Km[1, 2])
        cam.width, cam.height = int(width), int(height)
        cam.cam_id, cam.time = float(cam_id), float(0.0 if time is None else time)
        rd = self._hc.ray_dim
        rays = torch.empty((hi - lo, rd), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.hr_generate_rays(C.byref(cam), rd, lo, hi - lo, C.c_void_p(rays.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'hr_generate_rays')
        return rays

    def render_camera(self, pose, K, width, height, time=None, cam_id=0.0, pixel_range=None):
        """The viewer's frame path (utils/gui_utils.py:139-212, nlf/__init__.py:754-807) without
        its host round trips: pose -> rays -> rgb, all on the device and on the current stream."""
        return self.render(self.generate_rays(pose, K, width, height, time, cam_id, pixel_range), frame_time=time)['rgb']

    def pack_display(self, rgb, height, width, transpose=False, flip=False, rgba8=True):
        """The viewer's hand-over (utils/gui_utils.py:174-205) on the device: rgb (H*W, 3) as rendered -> the displayed
        buffer, transposed / flipped as NeRFGUI does on the host, as 8-bit RGBA (to8b, utils/__init__.py:47) or fp32 RGB."""
        import ctypes as C
        oh, ow = (width, height) if transpose else (height, width)
        rgb = rgb.contiguous().float()
        if rgb.device.type != 'cuda' or rgb.numel() != height * width * 3:
            raise ValueError('rgb must be a (H*W, 3) tensor on the HIP device')
        out = torch.empty((oh, ow, 4), dtype=torch.uint8, device=rgb.device) if rgba8 else \
            torch.empty((oh, ow, 3), dtype=torch.float32, device=rgb.device)
        with torch.cuda.device(rgb.device):
            _lib.check(_lib.load().hr_pack_display(C.c_void_p(rgb.data_ptr()), int(height), int(width), int(bool(transpose)), int(bool(flip)),
                                                   int(bool(rgba8)), C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(rgb.device).cuda_stream)), 'hr_pack_display')
        return out

    def forward_train(self, rays, white_bg=None, want_fields=False):
        """One differentiable forward of the training step (nlf/__init__.py:634-709 calls `self(coords)` in train mode):
        rgb (B, 3) WITHOUT the eval-mode clamp, with autograd history to the MLP, the planes / lines and basis_mat.
        white_bg: this step's background; default = the reference's draw `white_bg or rand() < 0.5` unless black_bg
        (tensorf_no_sample.py:236).  The activation schedules are the converged ones (see set_iter).
        want_fields: also return {'distances', 'points', 'render_weights', 'head'} of THIS forward (detached; hr_train_forward_fields)."""
        from . import train as T
        box = self.color_model.net.aabb
        if self._native is None or self._native_grid != self.grid_size or self._native_box != (box.data_ptr(), box._version):
            self.native()
        elif self._sched_built != self.cur_iter:
            self._sync_schedule()
        h, hc = self._native, self._hc
        rays = self._check_rays(rays)
        net_cfg = self.cfg['color']['net']
        if white_bg is None:
            white_bg = (bool(net_cfg.get('white_bg', False)) or bool(torch.rand(()) < 0.5)) and not bool(net_cfg.get('black_bg', False))
        types = [e['type'] for e in self.cfg['embedding']['embeddings'].values()]
        pred = self.embedding_model.embeddings[types.index('ray_prediction')]
        lvl0 = self._coarse_hc if self._coarse_hc is not None else hc       # the level the ray MLP belongs to
        if lvl0.mlp_layers == 0:                                # ZeroMLP, nlf/nets/mlp.py:14-33
            head = torch.zeros((rays.shape[0], lvl0.z_channels * lvl0.preds_per_z), dtype=torch.float32, device=rays.device)
        else:
            feats = T.ray_features(h, rays, lvl0.mlp_in)
            if self.train_fused_mlp and self._coarse_hc is None and lvl0.mlp_hidden == 256 and lvl0.mlp_layers >= 2 and not self.train_deterministic:
                # opt-in: one launch for the six layers (hr_mlp_train_forward: bf16 split arithmetic, head within 7e-6 of max |head|).  Not
                # the default: a pre-activation within that error of zero takes the other branch of the LeakyReLU, and the reference's
                # gradient (its autograd goldens, 1e-3 of a tensor's largest entry) is only met by the layer-by-layer forward, whose
                # GEMMs carry 24 mantissa bits (HipLinear); DESIGN 11
                head = T.mlp_forward_fused(h, rays, feats, pred.net, lvl0.mlp_skip_mask, lvl0.z_channels * lvl0.preds_per_z)
            else:
                head = T.mlp_forward(pred.net, feats, lvl0.mlp_skip_mask)
        if self._coarse_hc is not None:                         # point_prediction cascade (point.py:137-203)
            rows = T.CoarseRows.apply(h, rays, head, lvl0.z_channels, hc.casc_row_dim)
            point = self.embedding_model.embeddings[types.index('point_prediction')]
            head = T.mlp_forward(point.net, T.row_features(hc, rows), hc.mlp_skip_mask).reshape(rays.shape[0], -1)
        vm = self.color_model.net
        extra = ()
        if hc.color_table_views > 0:                            # ColorTransformEmbedding's table (point.py:558-602)
            extra = (self.embedding_model.embeddings[types.index('color_transform')].color_embedding,)
        # (the request travels as a class attribute: autograd.Function.apply takes tensors.  Whatever happens inside, neither the request nor a
        #  previous call's fields may outlive this call -- ADVICE r4)
        T.SampleStage.fields_out = None
        T.SampleStage.want_fields = hc.z_channels if want_fields else None
        try:
            rgb = T.SampleStage.apply(h, rays, head, white_bg, vm.basis_mat.weight, *T.grid_parameters(vm), *extra)
            f = T.SampleStage.fields_out
        finally:
            T.SampleStage.want_fields = None
            T.SampleStage.fields_out = None
        if want_fields:
            if f is None:
                raise RuntimeError('forward_train(want_fields=True): the sample stage returned no fields')
            f['head'] = head.detach()
            return rgb, f
        return rgb

    def forward(self, rays, render_kwargs=None):
        """LightfieldModel.forward (models.py:135-138).  In train mode with autograd enabled this is the
        differentiable path (forward_train); otherwise the inference renderer."""
        render_kwargs = render_kwargs or {}
        fields = list(render_kwargs.get('fields', []))
        if self.training and torch.is_grad_enabled() and not fields:
            return {'rgb': self.forward_train(rays)}
        if not fields:
            return {'rgb': self.render(rays)['rgb']}
        if self.training and torch.is_grad_enabled() and self._hc.z_channels <= 64:
            # INRSystem.training_step passes the regularizers' field list on the main forward (nlf/__init__.py:658-690): the colour
            # stays differentiable (training arithmetic: no eval-mode clamp, the per-step background draw) and the requested fields are
            # the per-sample values of the SAME forward pass (hr_train_forward_fields), detached -- no second, inference pass and no
            # re-upload of the weights.  A regulariser that needs d(field)/d(parameters) is outside the training path (SURVEY 8f-4
            # covers the colour loss and the plane regularisers).
            rgb, r = self.forward_train(rays, want_fields=True)
            r['rgb'] = rgb.detach()
            out = {k: v.detach() for k, v in self._forward_fields(rays, render_kwargs, r).items()}
            if not getattr(self, '_warned_detached_fields', False):
                import warnings
                self._warned_detached_fields = True
                warnings.warn('HipLightfieldModel.forward in train mode: the fields ' + ', '.join(sorted(k for k in out if k != 'rgb')) +
                              ' are DETACHED values of the training forward -- a regulariser that needs their gradient contributes none '
                              '(differentiable: rgb, and the plane regularisers of hyperreel_amd.train)', RuntimeWarning, stacklevel=2)
            out['rgb'] = rgb
            return out
        out = self._forward_fields(rays, render_kwargs)
        if self.training and torch.is_grad_enabled():
            # INRSystem.training_step passes the regularizers' field list on the main forward (nlf/__init__.py:634-709):
            # the colour must stay differentiable.  rgb comes from the training arithmetic (no eval-mode clamp, the
            # per-step background draw); the requested fields are the inference kernel's values, detached -- a
            # regulariser that needs d(field)/d(parameters) is outside the training path (SURVEY 8f-4 covers the colour loss
            # and the plane regularisers).
            out = {k: v.detach() for k, v in out.items()}
            if not getattr(self, '_warned_detached_fields', False):
                import warnings
                self._warned_detached_fields = True
                warnings.warn('HipLightfieldModel.forward in train mode: the fields ' + ', '.join(sorted(k for k in out if k != 'rgb')) +
                              ' come from the inference kernels and are DETACHED -- a regulariser that needs their gradient contributes none '
                              '(differentiable: rgb, and the plane regularisers of hyperreel_amd.train); every such step also re-uploads the '
                              'weights for the inference pass', RuntimeWarning, stacklevel=2)
            out['rgb'] = self.forward_train(rays)
        return out

    # -- diagnostics surface (visualizers only) ------------------------------------------
    def _head_fields(self, rays, head):
        hc = self._hc
        B, Z, P = rays.shape[0], hc.z_channels, hc.preds_per_z
        h = head.view(B, Z, P)
        out = {}

        def act(f, x):
            y = x * f.act.inner + f.act.shift
            if f.act.type == 1:
                y = torch.sigmoid(y)
            elif f.act.type == 2:
                y = torch.tanh(y)
            return y * f.act.outer + f.act.add

        for name, f in (('color_scale', hc.f_color_scale), ('color_shift', hc.f_color_shift)):
            if f.offset >= 0:
                out[name] = act(f, h[..., f.offset:f.offset + f.channels]).reshape(B, -1)
        return out

    def embed(self, rays, render_kwargs=None):
        """LightfieldModel.embed (models.py:131-133): the flattened fields handed to the
        colour net (extract_fields lists of the YAMLs).  Diagnostics path."""
        return self._embed_from(rays, self.render(rays, want=('distances', 'points', 'head')))

    def _embed_from(self, rays, r):
        rays = self._check_rays(rays)
        B, Z = rays.shape[0], self._hc.z_channels
        x = {'points': r['points'].reshape(B, -1), 'distances': r['distances'],
             'viewdirs': rays[:, None, 3:6].expand(B, Z, 3).reshape(B, -1),
             'weights': torch.ones_like(r['distances'])}
        x.update(self._head_fields(rays, r['head']))
        if self._hc.advect:
            t = rays[:, -1:]
            fac, inv = self._hc.flow_fac, self._hc.flow_inv_fac
            base = torch.round((t * fac).clamp(0.0, self._hc.flow_kmax) - 1e-5) * inv
            x['base_times'] = base.expand(B, Z).contiguous()
            x['time_offset'] = (t - base).expand(B, Z).contiguous()
            x['times'] = t.expand(B, Z).contiguous()
        return x

    def _forward_fields(self, rays, render_kwargs, r=None):
        """render_kwargs `fields` / `no_over_fields` (tensorf_no_sample.py:254-278).  r: per-sample values already at hand (the training
        forward's own); default: one pass of the inference kernels."""
        fields = list(render_kwargs.get('fields', []))
        no_over = list(render_kwargs.get('no_over_fields', []))
        if render_kwargs.get('pred_weights_fields'):
            raise NotImplementedError('pred_weights_fields')
        if r is None:
            r = self.render(rays, want=('distances', 'points', 'render_weights', 'head'))
        x = self._embed_from(rays, r)
        B, Z = r['render_weights'].shape
        out = {'rgb': r['rgb']}
        w = r['render_weights']
        for key in fields:
            if key == 'render_weights':
                out[key] = w
            elif key in x:
                v = x[key]
                out[key] = v.view(B, -1) if key in no_over else torch.sum(w[..., None] * v.view(B, Z, -1), -2)
        return out


model_dict = {'lightfield': HipLightfieldModel, 'lightfield_hip': HipLightfieldModel}
