"""C3D2 speaker-embedding network (`/root/reference/model.py:104-191`): the PyTorch module -- what checkpoints load
into and what training differentiates -- and `FusedEmbedder`, its inference form: seven hand-written MFMA entry points of
libsvk (`svk_c3d2_stage1`, `svk_c3d2_stage2`, `svk_c3d2_conv31`, `svk_c3d2_conv32t`, `svk_c3d2_conv41`: csrc/c3d2.hip, two-piece
f16 products; `svk_c3d2_conv42`, `svk_c3d2_fc5`: csrc/c3d2_tail.hip, f32).

Same constructor arguments, same sub-module names (a reference-format checkpoint's `state_dict` loads unchanged), same
`forward(x, development=True)`, `load_checkpoint(d)` and `create_Speaker_Model(u)` as the reference.  Input convention
`(batch, 1, 20, 80, 40)` (`/root/reference/utils.py:368-379`).

Which code runs a forward:
  * eval mode, input on the GPU, one-channel 20 x 80 x 40 cubes, no gradient asked of the input -- what
    `evaluation.py:67-84,113-121` and `model.py:188-191,374-388` do -- : the libsvk kernels.  A cube IS a feature
    matrix of 1 600 rows with crop starts 0, 80, 160 ...: `svk_c3d2_stage1` reads it as such, nothing is copied;
  * training mode, gradients, three-channel cubes (`utils.FeatureCube3C`), tensors on the host: the torch layers
    (autograd needs them; the north-star leaves the training forward on PyTorch-ROCm).
`model.inference_kernels = False` keeps an instance on the torch layers (A/B comparisons in tools).
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

# name suffix, in_ch (None = num_channels), out_ch, kernel, stride, maxpool after
_LAYERS = (
    ("1_1", None, 16, (3, 1, 5), (1, 1, 1), False),
    ("1_2", 16, 16, (3, 9, 1), (1, 2, 1), True),
    ("2_1", 16, 32, (3, 1, 4), (1, 1, 1), False),
    ("2_2", 32, 32, (3, 8, 1), (1, 2, 1), True),
    ("3_1", 32, 64, (3, 1, 3), (1, 1, 1), False),
    ("3_2", 64, 64, (3, 7, 1), (1, 1, 1), False),
    ("4_1", 64, 128, (3, 1, 3), (1, 1, 1), False),
    ("4_2", 128, 128, (3, 7, 1), (1, 1, 1), False),
)
EMBED_DIM = 128
_FLAT = 4 * 3 * 3 * 128
CUBE_SHAPE = (1, 20, 80, 40)          # channel, crops, frames, coefficients (utils.py:20-21, 368-379)
_EMBEDDERS = weakref.WeakKeyDictionary()   # model -> (state key, FusedEmbedder): kept off the module (deepcopy / pickling stay plain)


class C3D2(nn.Module):
    inference_kernels = True          # False on an instance: forward stays on the torch layers whatever the mode

    def __init__(self, n_labels, num_channels):
        super().__init__()
        self.n_labels, self.num_channels = n_labels, num_channels
        # creation order = the reference's, so that the same torch seed draws
        # the same initial weights (model.py:110-139)
        for tag, cin, cout, kernel, stride, pool in _LAYERS:
            cin = num_channels if cin is None else cin
            setattr(self, "conv" + tag, nn.Conv3d(cin, cout, kernel_size=kernel, stride=stride))
            setattr(self, "batch_norm" + tag, nn.BatchNorm3d(num_features=cout))
            setattr(self, "PReLu" + tag, nn.PReLU())
            if pool:
                setattr(self, "pool" + tag[0], nn.MaxPool3d(kernel_size=(1, 1, 2), stride=(1, 1, 2)))
        self.FC5 = nn.Linear(_FLAT, EMBED_DIM)
        self.PReLu5 = nn.PReLU()
        self.FC6 = nn.Linear(EMBED_DIM, n_labels)

    def torch_layers(self, x):
        """conv -> BatchNorm -> PReLU (-> pool) x 8 -> FC5 on torch operators (model.py:141-169): training, autograd,
        host tensors, three-channel cubes."""
        for tag, _, _, _, _, pool in _LAYERS:
            x = getattr(self, "conv" + tag)(x)
            x = getattr(self, "batch_norm" + tag)(x)
            x = getattr(self, "PReLu" + tag)(x)
            if pool:
                x = getattr(self, "pool" + tag[0])(x)
        return self.FC5(x.view(-1, _FLAT))

    def runs_on_kernels(self, x):
        """True when forward(x) is an inference call the libsvk network covers (see the module docstring)."""
        return bool(self.inference_kernels and not self.training and isinstance(x, torch.Tensor) and x.is_cuda
                    and x.dim() == 5 and tuple(x.shape[1:]) == CUBE_SHAPE and x.dtype == torch.float32
                    and self.num_channels == 1 and not (torch.is_grad_enabled() and x.requires_grad))

    def forward(self, x, development=True):
        if self.runs_on_kernels(x):
            x = self.fused_inference()(x)
        else:
            x = self.torch_layers(x)
        if development:
            x = F.softmax(self.FC6(self.PReLu5(x)), dim=1)
        return x

    def load_checkpoint(self, checkpoint_dict):
        """New model with `checkpoint_dict["state_dict"]` loaded; `module.`
        prefixes left by DataParallel are stripped (model.py:177-186)."""
        model = C3D2(n_labels=self.n_labels, num_channels=self.num_channels)
        if torch.cuda.is_available():
            model.cuda()
        wanted = model.state_dict()
        loaded = {}
        for key, value in checkpoint_dict["state_dict"].items():
            key = key.replace("module.", "")
            if key in wanted:
                loaded[key] = value
        model.load_state_dict(loaded)
        return model

    def create_Speaker_Model(self, utterance):
        self.eval()
        return self.forward(utterance, development=False)

    # ---- MI355X inference path -------------------------------------------
    def _state_key(self):
        """Identity and version of every tensor the embedding depends on: in-place updates (an optimiser step,
        load_state_dict) bump `_version`, `.to(device)` changes the storage."""
        return tuple((k, v.data_ptr(), v._version) for k, v in self.state_dict(keep_vars=True).items()
                     if not k.startswith(("FC6", "PReLu5")))

    def fused_inference(self):
        """The embedding-only inference form of the CURRENT weights (eval-mode BatchNorm folded into the convolutions,
        operands in the kernels' lane order): same maths as forward(development=False).  Cached until a weight changes."""
        key = self._state_key()
        hit = _EMBEDDERS.get(self)
        if hit is None or hit[0] != key:
            hit = _EMBEDDERS[self] = (key, FusedEmbedder(self))
        return hit[1]


class FusedEmbedder:
    """C3D2's forward(development=False) as seven libsvk kernels; the weights are a snapshot of the model at build time.
    Built for C3D2's layer shapes on one-channel cubes only: anything else raises (no framework fallback here -- the torch
    module itself is the path for other shapes)."""

    def __init__(self, model):
        self.device = model.conv1_1.weight.device       # tables are built where the weights live; kernels need the GPU
        self._eng = None
        self.stages = []
        self.act_scale = []     # per layer: the power of two per channel its output is carried in (all ones for an ordinary checkpoint)
        with torch.no_grad():
            s_in = None
            for tag, _, _, _, stride, pool in _LAYERS:
                conv = getattr(model, "conv" + tag)
                bn = getattr(model, "batch_norm" + tag)
                act = getattr(model, "PReLu" + tag)
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = conv.weight * scale.view(-1, 1, 1, 1, 1)
                b = (conv.bias - bn.running_mean) * scale + bn.bias
                # Activations travel as half pairs (include/svk.h): 22 bits while a value's pieces are normal halves, an absolute
                # floor of 2^-25 below 2^-3 and nothing above 65 504.  A network is the same function when channel c of a layer is
                # scaled by a > 0 (its BatchNorm's gamma and beta times a -- PReLU and the pools are positively homogeneous) and
                # the next layer's weights on that channel by 1 / a; the pieces are not.  So the channel's scale is FIXED here: its
                # activations are carried times 2^k with k = -round(log2(|gamma| + |beta|)) (the channel's spread and offset
                # behind BatchNorm), the next layer takes 2^-k into its weights -- exact in f32, k = 0 for every channel of a
                # checkpoint trained from PyTorch's initialisation, and FC5 takes the last layer's.
                # (round(log2 m) from m's own exponent and mantissa: the same k shift for m and 2^j m, whatever log2 rounds to)
                mant, expo = torch.frexp((bn.weight.abs() + bn.bias.abs()).float().clamp(min=2.0 ** -40, max=2.0 ** 40))
                k = -(expo - (mant < 0.5 ** 0.5).to(expo.dtype))
                s_out = torch.exp2(k.to(torch.float32))
                w = w * s_out.view(-1, 1, 1, 1, 1)
                b = b * s_out
                if s_in is not None:
                    w = w / s_in.view(1, -1, 1, 1, 1)
                self.stages.append((w.contiguous(), b.contiguous(), act.weight.detach().clone(), tuple(conv.stride), pool, False))
                self.act_scale.append(s_out)
                s_in = s_out
            self.fc_w = model.FC5.weight.detach().clone()
            if self.fc_w.shape[1] % s_in.numel() == 0:       # model.py:168 flattens NCDHW: column = channel * positions + position
                self.fc_w = (self.fc_w.view(self.fc_w.shape[0], s_in.numel(), -1) / s_in.view(1, -1, 1)).reshape(self.fc_w.shape).contiguous()
            self.fc_b = model.FC5.bias.detach().clone()
            for w, b, *_ in self.stages:
                if not (bool(torch.isfinite(w).all()) and bool(torch.isfinite(b).all()) and float(w.abs().max()) < 65504.0):
                    raise ValueError("a BatchNorm-folded weight of this checkpoint is not finite or exceeds 65 504: outside what "
                                     "libsvk's half-pair operands represent")
        tables = (self.stage1_tables(), self.stage2_tables(), self.conv31_tables(), self.conv32t_tables(),
                  self.conv41_tables(), self.conv42_tables(), self.fc5_tables())
        if any(t is None for t in tables):
            raise ValueError("libsvk's network kernels are built for C3D2's layers on one-channel 20 x 80 x 40 cubes "
                             "(model.py:110-139); this model's layers differ")
        self._starts = {}

    @property
    def eng(self):
        if self._eng is None:
            if self.device.type != "cuda":
                raise RuntimeError("FusedEmbedder runs on the GPU (libsvk); move the model to the device -- there is no CPU fallback")
            from .engine import get_engine
            self._eng = get_engine(self.device.index)
        return self._eng

    def crop_starts(self, n, device):
        """[n, 20] int32: 0, 80, 160 ... -- a cube read as 1 600 feature rows."""
        hit = self._starts.get(n)
        if hit is None:
            from . import constants as c
            row = torch.arange(c.CUBE_CROPS, dtype=torch.int32, device=device) * c.CUBE_FRAMES
            hit = self._starts[n] = row[None, :].expand(n, -1).contiguous()
            if len(self._starts) > 8:
                self._starts.pop(next(iter(self._starts)))
        return hit

    @torch.no_grad()
    def embed_features(self, feat, crop_idx, timed=None):
        """feature rows [n, T, 40] + crop starts [n, 20] -> embeddings [n, 128].  The cube (utils.py:351-379) is never
        materialised: the first-block kernel gathers its patches from the feature rows.  `timed(name, fn)`: the
        pipeline's HIP-event hook around each kernel (bench.py)."""
        run = timed or (lambda name, fn: fn())
        eng = self.eng
        y = run("stage1", lambda: eng.c3d2_stage1(feat, crop_idx, self.stage1_tables()))
        z = run("stage2", lambda: eng.c3d2_stage2(y, self.stage2_tables()))
        y = run("conv3_1", lambda: eng.c3d2_conv31(z, self.conv31_tables()))
        y = run("conv3_2", lambda: eng.c3d2_conv32t(y, self.conv32t_tables()))
        y = run("conv4_1", lambda: eng.c3d2_conv41(y, self.conv41_tables()))
        y = run("conv4_2", lambda: eng.c3d2_conv42(y, self.conv42_tables()))
        return run("fc5", lambda: eng.c3d2_fc5(y, self.fc5_tables()))

    @torch.no_grad()
    def __call__(self, cubes, batch=4096):
        """cubes [n, 1, 20, 80, 40] f32 on the device -> [n, 128]: the cube's memory is read as [n, 1 600, 40] feature rows
        with crop starts 0, 80, ... (a view: nothing is copied or re-gathered)."""
        if cubes.dim() != 5 or tuple(cubes.shape[1:]) != CUBE_SHAPE:
            raise ValueError("expected cubes of shape (n, 1, 20, 80, 40), got %s" % (tuple(cubes.shape),))
        x = self.eng.to_device(cubes, torch.float32)
        n = x.shape[0]
        if n:       # (this per-call surface only: the batched pipeline feeds its own features; aminmax: no temporary, NaN propagates)
            lo, hi = (float(v) for v in torch.aminmax(x))
        if n and not (lo > -65504.0 and hi < 65504.0):
            raise ValueError("cube values must be finite and below 65 504 in magnitude (the half-pair kernels' domain, include/svk.h); "
                             "the reference's features are log energies / MFCCs within +-100")
        rows = x.view(n, CUBE_SHAPE[1] * CUBE_SHAPE[2], CUBE_SHAPE[3])
        if n <= batch:
            return self.embed_features(rows, self.crop_starts(n, x.device)) if n else x.new_empty((0, EMBED_DIM))
        out = torch.empty((n, EMBED_DIM), dtype=torch.float32, device=x.device)
        for lo in range(0, n, batch):
            hi = min(n, lo + batch)
            out[lo:hi] = self.embed_features(rows[lo:hi], self.crop_starts(hi - lo, x.device))
        return out

    # ---- the first block (conv1_1 .. pool1) as one libsvk kernel ------------------------------------
    def stage1_tables(self):
        """Operand blocks of `svk_c3d2_stage1` (two-piece f16 products; include/svk.h): the BN-folded weights of conv1_1 /
        conv1_2 split into halves H = f16(w), L = f16(w - H) and laid out in the lane order of v_mfma_f32_16x16x32_f16's A
        operand (lane l = (co = l & 15, kk = l >> 4), eight halves: K = 8 kk + e), or None when the first block is not C3D2's
        (1 -> 16 k(3,1,5); 16 -> 16 k(3,9,1) stride (1,2,1); pool).
          w1blk [2][64][8]      : conv1_1, tap t = 8 (kk & 1) + e (t = 5 kd + kw; 15 -> 0): H for every kk | L for kk < 2, 0 above
          w2blk [14][2][64][8]  : conv1_2, tap pairs (a | b): ci = 8 (kk & 1) + e at tap a (kk < 2) / b (kk >= 2); H | L;
                                  pair 13 is tap (2, 8) alone against an [h | l] fragment: H at every kk | L for kk < 2, 0 above"""
        hit = getattr(self, "_stage1", False)
        if hit is not False:
            return hit
        self._stage1 = None
        (w1, b1, s1, st1, p1, _), (w2, b2, s2, st2, p2, _) = self.stages[0], self.stages[1]
        if (tuple(w1.shape) != (16, 1, 3, 1, 5) or tuple(w2.shape) != (16, 16, 3, 9, 1) or tuple(st1) != (1, 1, 1)
                or tuple(st2) != (1, 2, 1) or p1 or not p2):
            return None
        dev = w1.device

        def halves(w):
            h = w.to(torch.float16)
            return h, (w - h.to(torch.float32)).to(torch.float16)

        lane = torch.arange(64, device=dev)
        co, kk = lane & 15, lane >> 4
        e = torch.arange(8, device=dev)
        w1c = torch.cat([w1.contiguous().view(16, 15), torch.zeros((16, 1), device=dev)], 1)     # [co][t], t = 15: zero
        t = 8 * (kk & 1)[:, None] + e[None, :]                                                    # [64][8]
        h1, l1 = halves(w1c[co[:, None], t])
        w1blk = torch.stack([h1, torch.where((kk < 2)[:, None], l1, torch.zeros_like(l1))])       # [2][64][8]
        w2c = w2.contiguous()[:, :, :, :, 0]                                                      # [co][ci][kd][kh]
        pairs = [((p // 4, 2 * (p % 4)), (p // 4, 2 * (p % 4) + 1)) for p in range(12)] + [((0, 8), (1, 8)), ((2, 8), None)]
        ci = 8 * (kk & 1)[:, None] + e[None, :]
        w2blk = torch.zeros((14, 2, 64, 8), dtype=torch.float16, device=dev)
        for p, (ta, tb) in enumerate(pairs):
            wa = w2c[co[:, None], ci, ta[0], ta[1]]
            if tb is None:       # the last tap alone: the kernel reads it as [h | l], so H for every kk and L against the h half only
                h, l = halves(wa)
                l = torch.where((kk < 2)[:, None], l, torch.zeros_like(l))
            else:
                h, l = halves(torch.where((kk < 2)[:, None], wa, w2c[co[:, None], ci, tb[0], tb[1]]))
            w2blk[p, 0], w2blk[p, 1] = h, l
        slope01 = bool(((s1 >= 0) & (s1 <= 1)).all() and ((s2 >= 0) & (s2 <= 1)).all())   # one host read per checkpoint
        self._stage1 = (w1blk.contiguous(), b1.contiguous(),
                        s1.expand(16).contiguous() if s1.numel() == 1 else s1.contiguous(),
                        w2blk.contiguous(), b2.contiguous(),
                        s2.expand(16).contiguous() if s2.numel() == 1 else s2.contiguous(), slope01)
        return self._stage1

    def stage2_tables(self):
        """Operand fragments of `svk_c3d2_stage2` (conv2_1 16 -> 32 k(3,1,4); conv2_2 32 -> 32 k(3,8,1) stride
        (1,2,1) + pool), BN folded, or None when the layers differ:
          w21blk  [2 nt][6][2][64][8] halves: conv2_1 through two-piece f16 products (H | L blocks of tap pairs; see the code)
          w22blk  [2 nt][24][2][64][8] halves: conv2_2, one tap per K = 32 block (H | L)"""
        hit = getattr(self, "_stage2", False)
        if hit is not False:
            return hit
        self._stage2 = None
        (w1, b1, s1, st1, p1, _), (w2, b2, s2, st2, p2, _) = self.stages[2], self.stages[3]
        if (tuple(w1.shape) != (32, 16, 3, 1, 4) or tuple(w2.shape) != (32, 32, 3, 8, 1) or tuple(st1) != (1, 1, 1)
                or tuple(st2) != (1, 2, 1) or p1 or not p2):
            return None
        dev = w1.device
        lane = torch.arange(64, device=dev)
        ch, kq = lane & 15, lane >> 4
        a = w1.contiguous()[:, :, :, 0, :]                                   # [co][ci][kd][kw]
        # conv2_1 runs two-piece f16 products (see stage1_tables): [2 nt][6 pairs][2: H | L][64 lanes][8 halves], pair = 2 kd + kw / 2,
        # lane (co = 16 nt + (l & 15), kk = l >> 4): element e = W[co][ci = 8 (kk & 1) + e][kd][kw + (kk >= 2)]
        e8 = torch.arange(8, device=dev)
        ci8 = 8 * (kq & 1)[:, None] + e8[None, :]
        f21 = torch.empty((2, 6, 2, 64, 8), dtype=torch.float16, device=dev)
        for nt in range(2):
            for kd in range(3):
                for kw2 in range(2):
                    kw = 2 * kw2 + (kq >= 2).long()
                    w = a[(16 * nt + ch)[:, None], ci8, kd, kw[:, None]]
                    h = w.to(torch.float16)
                    f21[nt, 2 * kd + kw2, 0] = h
                    f21[nt, 2 * kd + kw2, 1] = (w - h.to(torch.float32)).to(torch.float16)
        bmat = w2.contiguous()[:, :, :, :, 0]                                # [co][ci][kd][kh]
        # conv2_2: [2 nt][24 taps][2: H | L][64 lanes][8 halves], tap = 8 kd + kh, element e = W[co][ci = 8 kk + e][kd][kh] (K = 32 = one tap)
        ci32 = 8 * kq[:, None] + e8[None, :]
        f22 = torch.empty((2, 24, 2, 64, 8), dtype=torch.float16, device=dev)
        for nt in range(2):
            for kd in range(3):
                for kh in range(8):
                    w = bmat[(16 * nt + ch)[:, None], ci32, kd, kh]
                    h = w.to(torch.float16)
                    f22[nt, 8 * kd + kh, 0] = h
                    f22[nt, 8 * kd + kh, 1] = (w - h.to(torch.float32)).to(torch.float16)

        def per_channel(t, n):
            return t.expand(n).contiguous() if t.numel() == 1 else t.contiguous()
        slope01 = bool(((s1 >= 0) & (s1 <= 1)).all() and ((s2 >= 0) & (s2 <= 1)).all())   # one host read per checkpoint
        self._stage2 = (f21.contiguous(), b1.contiguous(), per_channel(s1, 32), f22.contiguous(), b2.contiguous(),
                        per_channel(s2, 32), slope01)
        return self._stage2

    def conv31_tables(self):
        """Operand fragments of `svk_c3d2_conv31` (conv3_1: 32 -> 64, k(3,1,3), stride 1, no pool), BN folded, or None
        when the layer differs:  wblk [4 nt][9][2: H | L][64][8 halves]: lane (co = 16 nt + (l & 15), kk = l >> 4):
        W[co][8 kk + e][kd][kw], tap 3 kd + kw (two-piece f16 products, see stage1_tables)."""
        hit = getattr(self, "_conv31", False)
        if hit is not False:
            return hit
        self._conv31 = None
        if len(self.stages) < 5:
            return None
        w, b, sl, st, pool, _ = self.stages[4]
        if tuple(w.shape) != (64, 32, 3, 1, 3) or tuple(st) != (1, 1, 1) or pool:
            return None
        dev = w.device
        lane = torch.arange(64, device=dev)
        ch, kq = lane & 15, lane >> 4
        a = w.contiguous()[:, :, :, 0, :]                                    # [co][ci][kd][kw]
        ci = 8 * kq[:, None] + torch.arange(8, device=dev)[None, :]          # K = 32 = the 32 input channels of one tap
        frag = torch.empty((4, 9, 2, 64, 8), dtype=torch.float16, device=dev)
        for nt in range(4):
            for kd in range(3):
                for kw in range(3):
                    wv = a[(16 * nt + ch)[:, None], ci, kd, kw]
                    h = wv.to(torch.float16)
                    frag[nt, 3 * kd + kw, 0] = h
                    frag[nt, 3 * kd + kw, 1] = (wv - h.to(torch.float32)).to(torch.float16)
        slope = sl.expand(64).contiguous() if sl.numel() == 1 else sl.contiguous()
        self._conv31 = (frag.contiguous(), b.contiguous(), slope, bool(((sl >= 0) & (sl <= 1)).all()))
        return self._conv31

    @staticmethod
    def _depth_transformed(w):
        """Winograd F(2, 3) weight transform along depth of a BN-folded Conv3d weight [co][ci][3][kh][kw] ->
        [4 k][co][ci][kh][kw]: G0 = g0, G1 = ((g0 + g2) + g1) / 2, G2 = ((g0 + g2) - g1) / 2, G3 = g2 (f32, the same
        expressions the kernels of csrc/c3d2.hip evaluate in their prologues)."""
        g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
        return torch.stack((g0, 0.5 * ((g0 + g2) + g1), 0.5 * ((g0 + g2) - g1), g2))

    def _tail_conv_tables(self, li, shape, taps_axis):
        """Operand fragments of `svk_c3d2_conv42` from stage `li`, or None when the layer differs:
        wfrag [8 nt][chunks of 8 input channels][taps][4 k][64 lanes][2]: lane (co = 16 nt + (l & 15), kk = l >> 4),
        element e = G_k[co][8 chunk + 2 kk + e][tap]."""
        if len(self.stages) <= li:
            return None
        w, b, sl, st, pool, _ = self.stages[li]
        if tuple(w.shape) != shape or tuple(st) != (1, 1, 1) or pool:
            return None
        dev = w.device
        co, ci = shape[0], shape[1]
        g = self._depth_transformed(w.contiguous())                          # [4][co][ci][kh][kw]
        g = g[:, :, :, :, 0] if taps_axis == "h" else g[:, :, :, 0, :]        # [4][co][ci][taps]
        taps = g.shape[3]
        lane = torch.arange(64, device=dev)
        n_, kq = lane & 15, lane >> 4
        # frag[nt][chunk][tap][k][lane][e] = g[k][16 nt + n_][8 chunk + 2 kq + e][tap]
        frag = torch.empty((co // 16, ci // 8, taps, 4, 64, 2), dtype=torch.float32, device=dev)
        gg = g.view(4, co // 16, 16, ci // 8, 4, 2, taps)                    # [k][nt][n][chunk][kq][e][tap]
        frag.copy_(gg[:, :, n_, :, kq].permute(2, 3, 5, 1, 0, 4))            # advanced indices (n, kq) -> leading lane axis
        slope = sl.expand(co).contiguous() if sl.numel() == 1 else sl.contiguous()
        return (frag.contiguous(), b.contiguous(), slope, bool(((sl >= 0) & (sl <= 1)).all()))

    def conv32t_tables(self):
        """`svk_c3d2_conv32t` (conv3_2: 64 -> 64, k(3,7,1)), two-piece f16 products (see stage1_tables):
        wblk [4 nt][2 kb][21 taps][2: H | L][64 lanes][8 halves], element e = W[co = 16 nt + (l & 15)][32 kb + 8 kk + e][kd][kh],
        tap 7 kd + kh; bias, slope [64].  None when the layer differs."""
        hit = getattr(self, "_conv32t", False)
        if hit is not False:
            return hit
        self._conv32t = None
        if len(self.stages) < 6:
            return None
        w, b, sl, st, pool, _ = self.stages[5]
        if tuple(w.shape) != (64, 64, 3, 7, 1) or tuple(st) != (1, 1, 1) or pool:
            return None
        dev = w.device
        lane = torch.arange(64, device=dev)
        ch, kq = lane & 15, lane >> 4
        a = w.contiguous()[:, :, :, :, 0]                                    # [co][ci][kd][kh]
        e8 = torch.arange(8, device=dev)[None, :]
        blk = torch.empty((4, 2, 21, 2, 64, 8), dtype=torch.float16, device=dev)
        for nt in range(4):
            for kb in range(2):
                ci = 32 * kb + 8 * kq[:, None] + e8
                for kd in range(3):
                    for kh in range(7):
                        wv = a[(16 * nt + ch)[:, None], ci, kd, kh]
                        h = wv.to(torch.float16)
                        blk[nt, kb, 7 * kd + kh, 0] = h
                        blk[nt, kb, 7 * kd + kh, 1] = (wv - h.to(torch.float32)).to(torch.float16)
        slope = sl.expand(64).contiguous() if sl.numel() == 1 else sl.contiguous()
        self._conv32t = (blk.contiguous(), b.contiguous(), slope, bool(((sl >= 0) & (sl <= 1)).all()))
        return self._conv32t

    def conv41_tables(self):
        """`svk_c3d2_conv41` (conv4_1: 64 -> 128, k(3,1,3), stride 1, no pool), two-piece f16 products (see stage1_tables):
        wblk [8 nt][9 taps][2 kb][2: H | L][64 lanes][8 halves], element e = W[co = 16 nt + (l & 15)][32 kb + 8 kk + e][kd][kw],
        tap 3 kd + kw; bias, slope [128].  None when the layer differs."""
        hit = getattr(self, "_conv41", False)
        if hit is not False:
            return hit
        self._conv41 = None
        if len(self.stages) < 7:
            return None
        w, b, sl, st, pool, _ = self.stages[6]
        if tuple(w.shape) != (128, 64, 3, 1, 3) or tuple(st) != (1, 1, 1) or pool:
            return None
        dev = w.device
        lane = torch.arange(64, device=dev)
        ch, kq = lane & 15, lane >> 4
        a = w.contiguous()[:, :, :, 0, :]                                    # [co][ci][kd][kw]
        e8 = torch.arange(8, device=dev)[None, :]
        blk = torch.empty((8, 9, 2, 2, 64, 8), dtype=torch.float16, device=dev)
        for nt in range(8):
            for kb in range(2):
                ci = 32 * kb + 8 * kq[:, None] + e8
                for kd in range(3):
                    for kw in range(3):
                        wv = a[(16 * nt + ch)[:, None], ci, kd, kw]
                        h = wv.to(torch.float16)
                        blk[nt, 3 * kd + kw, kb, 0] = h
                        blk[nt, 3 * kd + kw, kb, 1] = (wv - h.to(torch.float32)).to(torch.float16)
        slope = sl.expand(128).contiguous() if sl.numel() == 1 else sl.contiguous()
        self._conv41 = (blk.contiguous(), b.contiguous(), slope, bool(((sl >= 0) & (sl <= 1)).all()))
        return self._conv41

    def conv42_tables(self):
        """`svk_c3d2_conv42` (conv4_2: 128 -> 128, k(3,7,1), stride 1, no pool)."""
        hit = getattr(self, "_conv42", False)
        if hit is False:
            hit = self._conv42 = self._tail_conv_tables(7, (128, 128, 3, 7, 1), "h")
        return hit

    def fc5_tables(self):
        """`svk_c3d2_fc5`: wfrag [4 d][8 nt][72 steps][64 lanes][4]: lane (j = 16 nt + (l & 15), kk = l >> 4), e:
        W5[j][c * 36 + d * 9 + pixel] for the K index 1 152 d + 16 step + 4 kk + e = ((d * 16 + chunk) * 9 + pixel) * 8 + c % 8
        (conv4_2's chunked output order; model.py:168 flattens NCDHW), and the bias.  None when FC5 is not 4 608 -> 128."""
        hit = getattr(self, "_fc5", False)
        if hit is not False:
            return hit
        self._fc5 = None
        if tuple(self.fc_w.shape) != (EMBED_DIM, _FLAT) or EMBED_DIM != 128:
            return None
        dev = self.fc_w.device
        # columns of the chunked order: [d][chunk][pixel][c8] -> torch column (8 chunk + c8) * 36 + d * 9 + pixel
        d, ch, px, c8 = torch.meshgrid(torch.arange(4, device=dev), torch.arange(16, device=dev),
                                       torch.arange(9, device=dev), torch.arange(8, device=dev), indexing="ij")
        col = ((8 * ch + c8) * 36 + d * 9 + px).reshape(-1)                  # [4608] in K order
        wk = self.fc_w[:, col]                                               # [128 j][4608 K]
        lane = torch.arange(64, device=dev)
        n_, kq = lane & 15, lane >> 4
        wv = wk.view(8, 16, 4, 72, 4, 4)                                     # [nt][n][d][step][kq][e]
        frag = wv[:, n_, :, :, kq].permute(2, 1, 3, 0, 4).contiguous()       # lane axis first -> [d][nt][step][lane][e]
        self._fc5 = (frag, self.fc_b.contiguous())
        return self._fc5

def seeded_model(seed, n_labels=1211, num_channels=1):
    """Random-init C3D2 under a fixed torch seed (the reference's checkpoint
    `Models/model_14_percent_best_so_far.pt` does not ship)."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = C3D2(n_labels, num_channels)
    torch.random.set_rng_state(gen_state)
    return model.eval()


def calibrate_batchnorm(model, cubes, batch=64):
    """Set every BatchNorm3d's running statistics to the statistics of `cubes` (cumulative
    average over batches), as a trained network's would be.  A fresh random-init C3D2 has
    running mean 0 / var 1 while its activations are far from that, which makes the 128-d
    output almost input-independent (all cosine scores within 1e-3 of 1.0) and the EER a coin
    flip decided by rounding noise; calibrated statistics give a well-conditioned, still
    untrained, embedding.  Deterministic given the weights and `cubes`."""
    norms = [m for m in model.modules() if isinstance(m, nn.BatchNorm3d)]
    saved = [(m.momentum, m.training) for m in norms]
    was_training = model.training
    for m in norms:
        m.reset_running_stats()
        m.momentum = None                      # cumulative moving average
    model.train()
    with torch.no_grad():
        for lo in range(0, cubes.shape[0], batch):
            model(cubes[lo:lo + batch], development=False)
    for m, (momentum, _) in zip(norms, saved):
        m.momentum = momentum
    model.train(was_training)
    return model


def perturb_inference_state(state_dict, seed):
    """Give BatchNorm running statistics, affine terms and PReLU slopes
    non-trivial values (a fresh init has mean 0 / var 1 / slope 0.25, which
    would leave BN folding and PReLU untested).  In place, deterministic."""
    gen = torch.Generator().manual_seed(seed)
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        if key.endswith("running_mean"):
            t.copy_(0.05 * torch.randn(t.shape, generator=gen))
        elif key.endswith("running_var"):
            t.copy_(0.5 + torch.rand(t.shape, generator=gen))
        elif "batch_norm" in key and key.endswith(".weight"):
            t.copy_(0.8 + 0.4 * torch.rand(t.shape, generator=gen))
        elif "batch_norm" in key and key.endswith(".bias"):
            t.copy_(0.05 * torch.randn(t.shape, generator=gen))
        elif "PReLu" in key:
            t.copy_(0.1 + 0.3 * torch.rand(t.shape, generator=gen))
    return state_dict


def _create_speaker_models_files():
    """model.py:351-388 as written: checkpoint, enrolment list, id table and WAV tree under
    `constants.ROOT` / `constants.DATA_ORIGIN`; one `{speaker_id}.pt` (a (1, 128) tensor) per
    speaker under ROOT/speaker_models, the LAST listed utterance winning (Q17)."""
    import os
    from . import constants as c
    from .evaluation import dataset_embeddings, load_indexed_labels
    from .utils import create_dataset
    model_path = os.path.join(c.ROOT, 'Models/model_14_percent_best_so_far.pt')
    save_speaker_models_path = os.path.join(c.ROOT, 'speaker_models')
    enrollment_set = os.path.join(c.ROOT, '50_first_ids.txt')
    indexed_labels = load_indexed_labels(c.ROOT + '/50_first_ids.npy')
    dataset = create_dataset(indexed_labels=indexed_labels, origin_file_path=enrollment_set)
    if not os.path.exists(save_speaker_models_path):
        os.mkdir(save_speaker_models_path)
    model = C3D2(100, 1).load_checkpoint(torch.load(model_path, map_location="cpu", weights_only=True))
    emb = dataset_embeddings(dataset, model).cpu()
    store = {}
    for i in range(len(dataset)):
        store[dataset.sound_files[i][0:7]] = emb[i:i + 1].clone()
    for sid, vec in store.items():
        torch.save(vec, '{}/{}.pt'.format(save_speaker_models_path, sid))
    return store


def create_speaker_models(model=None, cubes=None, speaker_ids=None, save_dir=None, batch=256):
    """Enrolment as `/root/reference/model.py:351-388` does it.  With no arguments: file-driven, the
    paths of `constants` (see `_create_speaker_models_files`).  With `(model, cubes, speaker_ids)`: the
    same on in-memory cubes.  Every utterance cube is embedded
    with `development=False`; the speaker model is the embedding of that speaker's LAST listed
    utterance -- the reference overwrites `{speaker_id}.pt` on each utterance, no averaging
    (Q17).  Returns `{speaker_id: (1, 128) CPU tensor}` and, with `save_dir`, writes the
    reference's `{speaker_id}.pt` files (readable by `evaluation.Evaluation`)."""
    import os
    if model is None and cubes is None:
        return _create_speaker_models_files()
    device = next(model.parameters()).device
    model.eval()
    store = {}
    with torch.no_grad():
        for lo in range(0, len(cubes), batch):
            x = torch.as_tensor(cubes[lo:lo + batch], dtype=torch.float32).to(device)
            emb = model(x, development=False).cpu()
            for k in range(emb.shape[0]):
                store[str(speaker_ids[lo + k])] = emb[k:k + 1].clone()
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        for sid, vec in store.items():
            torch.save(vec, os.path.join(save_dir, f"{sid}.pt"))
    return store
