"""C3D2 speaker-embedding network: the PyTorch module (what checkpoints load into, what training uses, the
parity oracle of the libsvk kernels) and `FusedEmbedder`, the inference path: since round 3 every layer runs in
libsvk (`svk_c3d2_stage1`, `svk_c3d2_stage2`, `svk_c3d2_conv31/32`: csrc/c3d2.hip; `svk_c3d2_conv41/42`, `svk_c3d2_fc5`:
csrc/c3d2_tail.hip); the same layers on PyTorch-ROCm (MIOpen / hipBLASLt, `svk_bias_prelu` behind each convolution)
remain as the A/B path behind the SVK_C3D2_* switches and for models whose layers are not C3D2's.

Mirrors `/root/reference/model.py:104-191`: same constructor arguments, same
sub-module names (so a reference-format checkpoint's `state_dict` loads
unchanged), same `forward(x, development=True)`, `load_checkpoint(d)` and
`create_Speaker_Model(u)`.  Input convention `(batch, 1, 20, 80, 40)`
(`/root/reference/utils.py:368-379`).

What is different, on purpose, for MI355X inference:
  * the network is described by one table and built in a loop;
  * `fused_inference()` folds eval-mode BatchNorm into the convolution weights
    and returns a lean callable for large-batch embedding extraction;
  * nothing here hard-codes `.cuda()` or prints (cf. Q20).
"""
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


class C3D2(nn.Module):
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

    def forward(self, x, development=True):
        for tag, _, _, _, _, pool in _LAYERS:
            x = getattr(self, "conv" + tag)(x)
            x = getattr(self, "batch_norm" + tag)(x)
            x = getattr(self, "PReLu" + tag)(x)
            if pool:
                x = getattr(self, "pool" + tag[0])(x)
        x = self.FC5(x.view(-1, _FLAT))
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
    def fused_inference(self, channels_last=False):
        """Embedding-only callable with BatchNorm folded into the convolutions
        (eval-mode statistics).  Same maths as forward(development=False)."""
        return FusedEmbedder(self, channels_last=channels_last)


class FusedEmbedder:
    """conv(+folded BN) -> PReLU chain; weights snapshot the model at build time."""

    def __init__(self, model, channels_last=False):
        self.channels_last = channels_last
        self.stages = []
        fmt = torch.channels_last_3d if channels_last else torch.contiguous_format
        with torch.no_grad():
            for tag, _, _, _, stride, pool in _LAYERS:
                conv = getattr(model, "conv" + tag)
                bn = getattr(model, "batch_norm" + tag)
                act = getattr(model, "PReLu" + tag)
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (conv.weight * scale.view(-1, 1, 1, 1, 1)).contiguous(memory_format=fmt)
                b = ((conv.bias - bn.running_mean) * scale + bn.bias).contiguous()
                # max-pool commutes with PReLU when the slope is >= 0 (PReLU is then non-decreasing):
                # pooling FIRST halves what the activation reads and writes (bit-identical result)
                pool_first = bool(pool and float(act.weight.detach().min()) >= 0.0)
                self.stages.append((w, b, act.weight.detach().clone(), stride, pool, pool_first))
            self.fc_w = model.FC5.weight.detach().clone()
            self.fc_b = model.FC5.bias.detach().clone()
            self.fc_w_cl = None
            # engine for svk_bias_prelu after conv3_1 .. conv4_2 (GPU + channels-last only; SVK_C3D2_TAIL=0 disables)
            self.fused_tail = None
            self.conv31_kernel = False       # conv3_1 in libsvk behind svk_c3d2_stage2 (SVK_C3D2_CONV31=0 disables)
            self.conv32_kernel = False       # ... and conv3_2 behind it (SVK_C3D2_CONV32=0 disables)
            self.conv4_kernels = False       # ... and conv4_1, conv4_2, FC5 (SVK_C3D2_CONV4=0 hands them back to PyTorch-ROCm)
            self.conv32t_kernel = False      # conv3_2 in the last block's shape (SVK_C3D2_CONV32T=0: the round-2 kernel)
            if channels_last and self.fc_w.is_cuda:
                import os
                if os.environ.get("SVK_C3D2_TAIL", "1") != "0":
                    from .engine import get_engine
                    self.fused_tail = get_engine(self.fc_w.device.index)
                    self.conv31_kernel = os.environ.get("SVK_C3D2_CONV31", "1") != "0"
                    self.conv32_kernel = self.conv31_kernel and os.environ.get("SVK_C3D2_CONV32", "1") != "0"
                    self.conv4_kernels = self.conv32_kernel and os.environ.get("SVK_C3D2_CONV4", "1") != "0"
                    self.conv32t_kernel = self.conv4_kernels and os.environ.get("SVK_C3D2_CONV32T", "1") != "0"
            # First layer as patch-matrix x weight GEMM.  MIOpen has no direct kernel for a 1-channel
            # Conv3d and falls back to im2col + per-group GEMM + layout transposes (6.5 ms per 978
            # cubes).  Here ONE strided copy gathers, for every group of G adjacent output columns, the
            # kd x (kw + G - 1) input window they share, and addmm multiplies it by a Toeplitz-expanded
            # weight matrix [kd (kw + G - 1), G * out_channels]: 3x the FLOPs of the plain patch matrix
            # but 1/4 of its bytes and a GEMM shape the library handles well (1.5 ms at G = 12); its
            # row-major result is already in channels-last (NDHWC) order.
            w0 = self.stages[0][0]
            self.first_as_gemm = bool(channels_last and w0.shape[1] == 1 and w0.shape[3] == 1 and
                                      self.stages[0][3] == (1, 1, 1))
            self._gemm_cache = {}
            self.row_fold = self._row_fold_tables(fmt) if channels_last else None

    def _row_fold_tables(self, fmt):
        """Stages 1-3 (conv1_2, conv2_1, conv2_2 of C3D2) with the PARITY OF THE ROW folded into the
        channels -- a pure re-indexing (same products, another summation order):
          * conv1_2 (kernel (kd, kh, 1), row stride 2) becomes a Toeplitz-widened conv that emits two
            output rows per position as 2 x co channels: kernel (kd, kh + 2, 1), row stride 4.  It has
            1.22x the multiply-adds but a GEMM N of 32 instead of 16: 5.3 -> 4.2 ms per 978 cubes;
          * conv2_1 (kernel (kd, 1, kw)) never mixes rows: the folded tensor's memory (.., w, parity, c) is
            read as 16 channels over the interleaved axis (w, parity) and the ORIGINAL weights run with a
            dilation of 2 along it (a view, no copy; 1.09 ms against 1.22 ms as a 2-group conv);
          * conv2_2 (kernel (kd, 8, 1), row stride 2) over rows 2 hp + parity is a stride-1 conv with
            kernel (kd, 4, 1) over row PAIRS whose input channels are (parity, ci): it un-folds the
            layout for free.
        The max-pool / PReLU between them act per channel along W and are unaffected (slopes tiled).
        Returns None when the layer shapes do not have this structure."""
        try:
            (w1, b1, s1, st1, _, _), (w2, b2, s2, st2, pool2, _), (w3, b3, s3, st3, _, _) = self.stages[1:4]
        except ValueError:
            return None
        ok = (tuple(st1) == (1, 2, 1) and w1.shape[4] == 1 and tuple(st2) == (1, 1, 1) and w2.shape[3] == 1 and
              not pool2 and
              tuple(st3) == (1, 2, 1) and w3.shape[4] == 1 and w3.shape[3] % 2 == 0 and
              w2.shape[1] == w1.shape[0] and w3.shape[1] == w2.shape[0])
        if not ok:
            return None
        co1, ci1, kd1, kh1, _ = w1.shape
        f1 = torch.zeros(2, co1, ci1, kd1, kh1 + 2, 1, dtype=w1.dtype, device=w1.device)
        for parity in range(2):
            f1[parity, :, :, :, 2 * parity:2 * parity + kh1, :] = w1
        f1 = f1.reshape(2 * co1, ci1, kd1, kh1 + 2, 1).contiguous(memory_format=fmt)
        f2 = w2.repeat(2, 1, 1, 1, 1).contiguous(memory_format=fmt)                  # groups = 2
        co3, ci3, kd3, kh3, _ = w3.shape
        # f3[co, parity * ci3 + ci, kd, khp] = w3[co, ci, kd, 2 khp + parity]
        f3 = (w3.reshape(co3, ci3, kd3, kh3 // 2, 2, 1).permute(0, 4, 1, 2, 3, 5)
              .reshape(co3, 2 * ci3, kd3, kh3 // 2, 1).contiguous(memory_format=fmt))
        def tile(slope):                      # nn.PReLU() has ONE slope; a per-channel one follows its channels
            return slope if slope.numel() == 1 else slope.repeat(2)
        return {"kh1": kh1, 1: (f1, b1.repeat(2), tile(s1), (1, 4, 1), 1),
                2: (f2, b2.repeat(2), tile(s2), (1, 1, 1), 2), 3: (f3, b3, s3, (1, 1, 1), 1)}

    def _first_layer_tables(self, ow):
        """(G, Toeplitz weight matrix, tiled bias) for an output width `ow`; G = largest divisor <= 12."""
        hit = self._gemm_cache.get(ow)
        if hit is None:
            w, b = self.stages[0][0], self.stages[0][1]
            co, kd, kw = w.shape[0], w.shape[2], w.shape[4]
            G = max(g for g in range(1, 13) if ow % g == 0)
            win = kw + G - 1
            wt = torch.zeros(kd, win, G, co, dtype=w.dtype, device=w.device)
            taps = w[:, 0, :, 0, :].permute(1, 2, 0)                      # (kd, kw, co)
            for g in range(G):
                wt[:, g:g + kw, g, :] = taps
            hit = (G, wt.reshape(kd * win, G * co).contiguous(), b.repeat(G))
            self._gemm_cache[ow] = hit
        return hit

    def first_layer_windows(self, n_crops, n_cols):
        """(kd, kw, G) of the patch matrix the first layer's GEMM consumes for cubes of `n_crops` x `n_cols`
        (svk_cube_gather_windows can write it directly), or None when the first layer is a plain conv."""
        if not self.first_as_gemm:
            return None
        w = self.stages[0][0]
        kd, kw = int(w.shape[2]), int(w.shape[4])
        G = self._first_layer_tables(n_cols - kw + 1)[0]
        return kd, kw, G

    @torch.no_grad()
    def from_windows(self, windows, n, n_crops, crop_frames, n_cols):
        """Embeddings from the first layer's patch matrix (see first_layer_windows) instead of the cube."""
        w = self.stages[0][0]
        kd, kw = int(w.shape[2]), int(w.shape[4])
        od, ow = n_crops - kd + 1, n_cols - kw + 1
        _, wt, bt = self._first_layer_tables(ow)
        x = torch.addmm(bt, windows, wt)
        x = x.view(n, od, crop_frames, ow, w.shape[0]).permute(0, 4, 1, 2, 3)     # NDHWC memory = channels_last_3d
        return self._run(x, first_done=True)

    @torch.no_grad()
    def __call__(self, x):
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last_3d)
        return self._run(x, first_done=False)

    def _channel_slopes(self, li):
        """PReLU slope of stage `li` as one value per output channel (nn.PReLU() holds a single one), cached."""
        cache = self.__dict__.setdefault("_slopes", {})
        if li not in cache:
            w, _, slope = self.stages[li][:3]
            cache[li] = slope.expand(w.shape[0]).contiguous() if slope.numel() == 1 else slope.contiguous()
        return cache[li]

    # ---- the first block (conv1_1 .. pool1) as one libsvk kernel ------------------------------------
    def stage1_tables(self):
        """Operand fragments of `svk_c3d2_stage1` (csrc/c3d2.hip) from the BN-folded weights of conv1_1 / conv1_2,
        or None when the first block is not C3D2's (1 -> 16 k(3,1,5); 16 -> 16 k(3,9,1) stride (1,2,1); pool).
          w1frag [4][64]   : lane (channel = l & 15, kq = l >> 4), GEMM row k = 4 jj + kq: tap (k // 5, k % 5) of
                             conv1_1, row 15 = 0 (padding); its bias goes separately
          w2frag [27][64][4]: lane (co = l & 15, kk = l >> 4), element e = W2[co][4 kk + e][kd][kh], tap = 9 kd + kh"""
        hit = getattr(self, "_stage1", False)
        if hit is not False:
            return hit
        self._stage1 = None
        (w1, b1, s1, st1, p1, _), (w2, b2, s2, st2, p2, _) = self.stages[0], self.stages[1]
        if (tuple(w1.shape) != (16, 1, 3, 1, 5) or tuple(w2.shape) != (16, 16, 3, 9, 1) or tuple(st1) != (1, 1, 1)
                or tuple(st2) != (1, 2, 1) or p1 or not p2):
            return None
        dev = w1.device
        lane = torch.arange(64, device=dev)
        ch, kq = lane & 15, lane >> 4
        w1frag = torch.empty((4, 64), dtype=torch.float32, device=dev)
        w1c = w1.contiguous().view(16, 15)                                  # [co][kd * 5 + kw]
        for jj in range(4):
            k = 4 * jj + kq
            w1frag[jj] = torch.where(k < 15, w1c[ch, k.clamp(max=14)], torch.zeros_like(b1[ch]))
        w2c = w2.contiguous()[:, :, :, :, 0]                                # [co][ci][kd][kh]
        w2frag = torch.empty((27, 64, 4), dtype=torch.float32, device=dev)
        for kd in range(3):
            for kh in range(9):
                for e in range(4):
                    w2frag[9 * kd + kh, :, e] = w2c[ch, 4 * kq + e, kd, kh]
        slope01 = bool(((s1 >= 0) & (s1 <= 1)).all() and ((s2 >= 0) & (s2 <= 1)).all())   # one host read per checkpoint
        self._stage1 = (w1frag.contiguous(), b1.contiguous(),
                        s1.expand(16).contiguous() if s1.numel() == 1 else s1.contiguous(),
                        w2frag.contiguous(), b2.contiguous(),
                        s2.expand(16).contiguous() if s2.numel() == 1 else s2.contiguous(), slope01)
        return self._stage1

    def stage2_tables(self):
        """Operand fragments of `svk_c3d2_stage2` (conv2_1 16 -> 32 k(3,1,4); conv2_2 32 -> 32 k(3,8,1) stride
        (1,2,1) + pool), BN folded, or None when the layers differ:
          w21frag [2 nt][12][64][4]   : lane (co = 16 nt + (l & 15), kk = l >> 4): W[co][4 kk + e][kd][kw], tap 4 kd + kw
          w22frag [2 nt][24][2][64][4]: W[co][16 chunk + 4 kk + e][kd][kh], tap 8 kd + kh"""
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
        f21 = torch.empty((2, 12, 64, 4), dtype=torch.float32, device=dev)
        for nt in range(2):
            for kd in range(3):
                for kw in range(4):
                    for e in range(4):
                        f21[nt, 4 * kd + kw, :, e] = a[16 * nt + ch, 4 * kq + e, kd, kw]
        bmat = w2.contiguous()[:, :, :, :, 0]                                # [co][ci][kd][kh]
        f22 = torch.empty((2, 24, 2, 64, 4), dtype=torch.float32, device=dev)
        for nt in range(2):
            for kd in range(3):
                for kh in range(8):
                    for chunk in range(2):
                        for e in range(4):
                            f22[nt, 8 * kd + kh, chunk, :, e] = bmat[16 * nt + ch, 16 * chunk + 4 * kq + e, kd, kh]

        def per_channel(t, n):
            return t.expand(n).contiguous() if t.numel() == 1 else t.contiguous()
        slope01 = bool(((s1 >= 0) & (s1 <= 1)).all() and ((s2 >= 0) & (s2 <= 1)).all())   # one host read per checkpoint
        self._stage2 = (f21.contiguous(), b1.contiguous(), per_channel(s1, 32), f22.contiguous(), b2.contiguous(),
                        per_channel(s2, 32), slope01)
        return self._stage2

    def conv31_tables(self):
        """Operand fragments of `svk_c3d2_conv31` (conv3_1: 32 -> 64, k(3,1,3), stride 1, no pool), BN folded, or None
        when the layer differs:  wfrag [4 nt][9][2][64][4]: lane (co = 16 nt + (l & 15), kk = l >> 4):
        W[co][16 chunk + 4 kk + e][kd][kw], tap 3 kd + kw."""
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
        frag = torch.empty((4, 9, 2, 64, 4), dtype=torch.float32, device=dev)
        for nt in range(4):
            for kd in range(3):
                for kw in range(3):
                    for chunk in range(2):
                        for e in range(4):
                            frag[nt, 3 * kd + kw, chunk, :, e] = a[16 * nt + ch, 16 * chunk + 4 * kq + e, kd, kw]
        slope = sl.expand(64).contiguous() if sl.numel() == 1 else sl.contiguous()
        self._conv31 = (frag.contiguous(), b.contiguous(), slope, bool(((sl >= 0) & (sl <= 1)).all()))
        return self._conv31

    def conv32_tables(self):
        """Operand fragments of `svk_c3d2_conv32` (conv3_2: 64 -> 64, k(3,7,1), stride 1, no pool), BN folded, or None
        when the layer differs:  wfrag [4 nt][21][4 chunks][64][4]: lane (co = 16 nt + (l & 15), kk = l >> 4):
        W[co][16 chunk + 4 kk + e][kd][kh], tap 7 kd + kh."""
        hit = getattr(self, "_conv32", False)
        if hit is not False:
            return hit
        self._conv32 = None
        if len(self.stages) < 6:
            return None
        w, b, sl, st, pool, _ = self.stages[5]
        if tuple(w.shape) != (64, 64, 3, 7, 1) or tuple(st) != (1, 1, 1) or pool:
            return None
        dev = w.device
        lane = torch.arange(64, device=dev)
        ch, kq = lane & 15, lane >> 4
        a = w.contiguous()[:, :, :, :, 0]                                    # [co][ci][kd][kh]
        frag = torch.empty((4, 21, 4, 64, 4), dtype=torch.float32, device=dev)
        for nt in range(4):
            for kd in range(3):
                for kh in range(7):
                    for chunk in range(4):
                        for e in range(4):
                            frag[nt, 7 * kd + kh, chunk, :, e] = a[16 * nt + ch, 16 * chunk + 4 * kq + e, kd, kh]
        slope = sl.expand(64).contiguous() if sl.numel() == 1 else sl.contiguous()
        self._conv32 = (frag.contiguous(), b.contiguous(), slope, bool(((sl >= 0) & (sl <= 1)).all()))
        return self._conv32

    @staticmethod
    def _depth_transformed(w):
        """Winograd F(2, 3) weight transform along depth of a BN-folded Conv3d weight [co][ci][3][kh][kw] ->
        [4 k][co][ci][kh][kw]: G0 = g0, G1 = ((g0 + g2) + g1) / 2, G2 = ((g0 + g2) - g1) / 2, G3 = g2 (f32, the same
        expressions the kernels of csrc/c3d2.hip evaluate in their prologues)."""
        g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
        return torch.stack((g0, 0.5 * ((g0 + g2) + g1), 0.5 * ((g0 + g2) - g1), g2))

    def _tail_conv_tables(self, li, shape, taps_axis):
        """Operand fragments of `svk_c3d2_conv41` / `svk_c3d2_conv42` from stage `li`, or None when the layer differs:
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
        """`svk_c3d2_conv32t` (conv3_2: 64 -> 64, k(3,7,1)) in the last block's operand format (host-transformed)."""
        hit = getattr(self, "_conv32t", False)
        if hit is False:
            hit = self._conv32t = self._tail_conv_tables(5, (64, 64, 3, 7, 1), "h")
        return hit

    def conv41_tables(self):
        """`svk_c3d2_conv41` (conv4_1: 64 -> 128, k(3,1,3), stride 1, no pool), BN folded, depth-transformed by the host."""
        hit = getattr(self, "_conv41", False)
        if hit is False:
            hit = self._conv41 = self._tail_conv_tables(6, (128, 64, 3, 1, 3), "w")
        return hit

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

    def tail_in_libsvk(self):
        """True when conv3_1 .. FC5 all have libsvk kernels for this model (then `from_stage2` calls no framework op)."""
        return bool(self.fused_tail is not None and self.conv31_kernel and self.conv32_kernel and self.conv4_kernels
                    and self.conv31_tables() is not None and self.conv32_tables() is not None
                    and self.conv41_tables() is not None and self.conv42_tables() is not None
                    and self.fc5_tables() is not None)

    @torch.no_grad()
    def tail_from_conv32(self, y, n, timed=None):
        """conv4_1 -> conv4_2 -> FC5 in libsvk from conv3_2's CHUNKED output [n][8][8][45][8] (csrc/c3d2_tail.hip)."""
        timed = timed or (lambda name, fn: fn())
        y = timed("conv4_1", lambda: self.fused_tail.c3d2_conv41(y, self.conv41_tables()))
        y2 = timed("conv4_2", lambda: self.fused_tail.c3d2_conv42(y, self.conv42_tables()))
        return timed("fc5", lambda: self.fused_tail.c3d2_fc5(y2, self.fc5_tables()))

    @torch.no_grad()
    def from_stage2(self, z, n, timed=None):
        """Embeddings from the output of `svk_c3d2_stage2`: the activation after pool2, [n][12][15][7][32].  conv3_1 ..
        FC5 run in libsvk too when the engine is there (SVK_C3D2_CONV31 / CONV32 / CONV4 = 0 hand layers back to
        PyTorch-ROCm).  `timed(name, fn)`: the pipeline's HIP-event hook around each kernel (bench.py)."""
        t31 = self.conv31_tables() if (self.fused_tail is not None and self.conv31_kernel and z.is_cuda) else None
        if t31 is not None:
            timed_ = timed or (lambda name, fn: fn())
            if self.conv32t_kernel and self.tail_in_libsvk() and self.conv32t_tables() is not None:
                # conv3_2 in the last block's shape (the default): conv3_1 writes the chunked, column-major layout it stages from
                y = timed_("conv3_1", lambda: self.fused_tail.c3d2_conv31(z.view(n, 12, 15, 7, 32), t31, chunked=True))
                yc = timed_("conv3_2", lambda: self.fused_tail.c3d2_conv32t(y, self.conv32t_tables()))
                return self.tail_from_conv32(yc, n, timed)
            y = timed_("conv3_1", lambda: self.fused_tail.c3d2_conv31(z.view(n, 12, 15, 7, 32), t31))
            t32 = self.conv32_tables() if self.conv32_kernel else None
            if t32 is not None:
                if (self.conv4_kernels and self.conv41_tables() is not None and self.conv42_tables() is not None
                        and self.fc5_tables() is not None):
                    # the whole rest of the network in libsvk: conv3_2 writes the chunked layout conv4_1 stages from
                    yc = timed_("conv3_2", lambda: self.fused_tail.c3d2_conv32(y, t32, chunked=True))
                    return self.tail_from_conv32(yc, n, timed)
                y = self.fused_tail.c3d2_conv32(y, t32)
                x = y.view(n, 8, 9, 5, 64).permute(0, 4, 1, 2, 3)           # (n, 64, 8, 9, 5), channels_last_3d memory
                return self._run(x, start=6)
            x = y.view(n, 10, 15, 5, 64).permute(0, 4, 1, 2, 3)             # (n, 64, 10, 15, 5), channels_last_3d memory
            return self._run(x, start=5)
        x = z.view(n, 12, 15, 7, 32).permute(0, 4, 1, 2, 3)                 # (n, 32, 12, 15, 7), channels_last_3d memory
        return self._run(x, start=4)

    @torch.no_grad()
    def from_stage1(self, y, n):
        """Embeddings from the output of `svk_c3d2_stage1` -- the activation after pool1, in the row-folded
        channels-last layout [n][16][18][18][2][16] when `self.row_fold` exists, plain [n][16][36][18][16] otherwise."""
        if self.row_fold is not None:
            x = y.view(n, 16, 18, 18, 32).permute(0, 4, 1, 2, 3)           # (n, 32, 16, 18, 18), channels_last_3d memory
            return self._run(x, start=2, fold=self.row_fold)
        x = y.view(n, 16, 36, 18, 16).permute(0, 4, 1, 2, 3)
        return self._run(x, start=2)

    def _run(self, x, first_done=False, start=0, fold=None):
        for li, (w, b, slope, stride, pool, pool_first) in enumerate(self.stages):
            if li < start:
                continue
            groups = 1
            if li == 1 and self.row_fold is not None:
                # rows fold only when conv1_2's output has an even number of rows that the stride-4 form reproduces
                h_out = (x.shape[3] - self.row_fold["kh1"]) // 2 + 1
                if h_out % 2 == 0 and (x.shape[3] - self.row_fold["kh1"] - 2) // 4 + 1 == h_out // 2:
                    fold = self.row_fold
            if fold is not None and li == 2 and x.is_contiguous(memory_format=torch.channels_last_3d):
                # (parity, c) channels over w  ==  c channels over the interleaved (w, parity) axis: same bytes
                n_, c2, d_, hp_, w_ = x.shape
                xv = x.as_strided((n_, c2 // 2, d_, hp_, 2 * w_), (x.stride(0), 1, x.stride(2), x.stride(3), c2 // 2))
                y = F.prelu(F.conv3d(xv, w, b, dilation=(1, 1, 2)), slope)
                # the view below re-reads y's memory as NDHWC; a convolution solver may hand back NCDHW
                y = y.contiguous(memory_format=torch.channels_last_3d)
                co = y.shape[1]
                x = y.as_strided((n_, 2 * co, y.shape[2], hp_, y.shape[4] // 2),
                                 (y.stride(0), 1, y.stride(2), y.stride(3), 2 * co))
                continue
            if fold is not None and li in (1, 2, 3):
                w, b, slope, stride, groups = fold[li]
            if li == 0 and first_done:
                pass
            elif li == 0 and self.first_as_gemm:
                n, _, d, h, wd = x.shape
                kd, kw = w.shape[2], w.shape[4]
                od, ow = d - kd + 1, wd - kw + 1
                G, wt, bt = self._first_layer_tables(ow)
                xs = x.reshape(n, d, h, wd)
                windows = xs.as_strided((n, od, h, ow // G, kd, kw + G - 1), (d * h * wd, h * wd, wd, G, h * wd, 1))
                x = torch.addmm(bt, windows.reshape(n * od * h * (ow // G), kd * (kw + G - 1)), wt)
                x = x.view(n, od, h, ow, w.shape[0]).permute(0, 4, 1, 2, 3)     # NDHWC memory = channels_last_3d
            elif (self.fused_tail is not None and not pool and groups == 1 and w.shape[0] % 4 == 0 and x.is_cuda
                  and x.is_contiguous(memory_format=torch.channels_last_3d)):
                # conv3_1 .. conv4_2: the framework's convolution WITHOUT bias, then bias + PReLU in one in-place
                # libsvk pass (svk_bias_prelu) instead of a bias-add kernel and a PReLU kernel
                y = F.conv3d(x, w, None, stride=stride).contiguous(memory_format=torch.channels_last_3d)
                x = self.fused_tail.bias_prelu_(y, b, self._channel_slopes(li))
                continue
            else:
                x = F.conv3d(x, w, b, stride=stride, groups=groups)
            if pool_first:
                # MaxPool3d((1,1,2)) as one element-wise max of the even and odd columns (an odd last
                # column is dropped, as the pooling floor does): 2.3x faster than max_pool3d here
                w2 = x.shape[-1] // 2 * 2
                x = F.prelu(torch.maximum(x[..., 0:w2:2], x[..., 1:w2:2]), slope)
            else:
                x = F.prelu(x, slope)
                if pool:
                    x = F.max_pool3d(x, kernel_size=(1, 1, 2), stride=(1, 1, 2))
        if x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous():
            # flatten in memory order (d, h, w, c) -- a view -- against FC5's columns permuted to match
            # (model.py:168 flattens NCDHW: column c * 36 + (d, h, w))
            if self.fc_w_cl is None:
                cdhw = self.fc_w.view(self.fc_w.shape[0], x.shape[1], -1)                    # [out][c][dhw]
                self.fc_w_cl = cdhw.permute(0, 2, 1).reshape(self.fc_w.shape[0], _FLAT).contiguous()
            return F.linear(x.permute(0, 2, 3, 4, 1).reshape(x.shape[0], _FLAT), self.fc_w_cl, self.fc_b)
        return F.linear(x.reshape(x.shape[0], _FLAT), self.fc_w, self.fc_b)


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
