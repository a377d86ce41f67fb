#!/usr/bin/env python3
"""The reference's file-driven entry points, as a maintainer would call them after swapping the imports (INTEGRATION.md
section 2): a synthetic tree in the reference's layout (synth.write_verification_tree: checkpoint, id list, id table, WAVs),
then `model.create_speaker_models()` and `evaluation.evaluate()` with no arguments (/root/reference/model.py:351-388,
evaluation.py:90-146).  Run under `rocprofv3 --kernel-trace --stats` (tools/refresh_profiles.sh) its kernel table shows
what those calls execute: libsvk kernels only -- no `ck::`, `naive_conv`, MIOpen or `at::native::*conv*` row.
The tree carries the committed trained checkpoint and `constants.NORMALIZE` is switched on (the checkpoint was trained on
CMVN-normalised features), so the printed EER / accuracy are those of a trained network on speakers it never saw.
    python tools/profile_evaluate.py [n_speakers] [utts_per_speaker]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                            # noqa: E402

from speaker_verification_amd import constants, evaluation, model, synth     # noqa: E402

n_spk = int(sys.argv[1]) if len(sys.argv) > 1 else 12
per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
root = tempfile.mkdtemp(prefix="svk_eval_tree_")
ckpt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speaker_verification_amd", "checkpoints", "c3d2_synth.pt")
data_dir, rel, _ = synth.write_verification_tree(root, n_speakers=n_spk, utts_per_speaker=per, n_samples=48000, checkpoint=ckpt)
constants.ROOT, constants.DATA_ORIGIN, constants.NORMALIZE = root, data_dir, True
os.chdir(root)                                                                # eer_auc.png lands in the CWD, as in the reference
np.random.seed(1)
store = model.create_speaker_models()
np.random.seed(2)
res = evaluation.evaluate()
print("enrolled %d speakers from %d files; evaluate(): EER %.4f AUC %.4f accuracy %.1f %%"
      % (len(store), len(rel), res["eer"], res["auc"], res["accuracy"]), file=sys.stderr)
