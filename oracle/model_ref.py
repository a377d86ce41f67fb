"""CPU oracle for the embedding network and the cube transform -- TEST
INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, on torch-CPU float32 and as pure functions of a `state_dict`:
  * `C3D2.forward(x, development=False)`  (`/root/reference/model.py:141-175`,
    layer shapes from `model.py:110-139`): 8 x (Conv3d -> BatchNorm3d(eval) ->
    PReLU), two MaxPool3d, flatten 4*3*3*128 -> FC5 -> 128-d embedding;
  * `FeatureCube((80, 40, 20))`  (`/root/reference/utils.py:351-379`, Q15).

Parity status: PINNED by `tests/golden/c3d2_embed.npz`, produced by
`tools/make_golden.py` instantiating the reference's own `C3D2` under a fixed
torch seed and running it on a fixed cube.
"""
import numpy as np
import torch
import torch.nn.functional as F

# (conv name, bn name, prelu name, stride, pool-after?) in forward order.
# Kernel sizes live in the weights themselves.  model.py:110-139.
_BLOCKS = (
    ("conv1_1", "batch_norm1_1", "PReLu1_1", (1, 1, 1), False),
    ("conv1_2", "batch_norm1_2", "PReLu1_2", (1, 2, 1), True),
    ("conv2_1", "batch_norm2_1", "PReLu2_1", (1, 1, 1), False),
    ("conv2_2", "batch_norm2_2", "PReLu2_2", (1, 2, 1), True),
    ("conv3_1", "batch_norm3_1", "PReLu3_1", (1, 1, 1), False),
    ("conv3_2", "batch_norm3_2", "PReLu3_2", (1, 1, 1), False),
    ("conv4_1", "batch_norm4_1", "PReLu4_1", (1, 1, 1), False),
    ("conv4_2", "batch_norm4_2", "PReLu4_2", (1, 1, 1), False),
)


def c3d2_embed(state, x, bn_eps=1e-5):
    """128-d embeddings for cubes x of shape (B, 1, 20, 80, 40), float32.
    Eval-mode batch norm (running statistics).  model.py:141-170."""
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        for conv, bn, act, stride, pool in _BLOCKS:
            x = F.conv3d(x, state[conv + ".weight"], state[conv + ".bias"], stride=stride)
            x = F.batch_norm(x, state[bn + ".running_mean"], state[bn + ".running_var"],
                             state[bn + ".weight"], state[bn + ".bias"],
                             training=False, eps=bn_eps)
            x = F.prelu(x, state[act + ".weight"])
            if pool:
                x = F.max_pool3d(x, kernel_size=(1, 1, 2), stride=(1, 1, 2))  # model.py:117,124
        x = x.reshape(-1, 4 * 3 * 3 * 128)                                     # model.py:168
        x = F.linear(x, state["FC5.weight"], state["FC5.bias"])                # model.py:169
    return x


def feature_cube(feature, crop_idx, num_frames=80):
    """(T, C) features + explicit crop starts -> (1, n_crops, num_frames, C)
    float32: crop u is feature[idx[u]:idx[u]+num_frames].  utils.py:364-379.
    The reference draws idx = np.random.randint(T - num_frames, size=20) from
    the global RNG (utils.py:372); here it is an input so both sides agree."""
    feature = np.asarray(feature)
    cube = np.zeros((len(crop_idx), num_frames, feature.shape[1]), dtype=np.float32)
    for u, start in enumerate(crop_idx):
        cube[u] = feature[start:start + num_frames, :]
    return cube[None]


def draw_crops(rng, n_feature_frames, num_frames=80, n_crops=20):
    """The reference's draw: `randint(T - num_frames, size=n_crops)` on a
    legacy RandomState (utils.py:15,372)."""
    return rng.randint(n_feature_frames - num_frames, size=n_crops)
