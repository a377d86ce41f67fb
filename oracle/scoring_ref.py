"""CPU oracle for scoring, EER/AUC and the Siamese loss -- TEST INFRASTRUCTURE,
NOT PRODUCT CODE.

Restates (paths relative to `/root/reference/`):
  * `Evaluation.compute_Similarity`  (`evaluation.py:67-84`, Q18): float32
    cosine of the utterance embedding against each enrolled speaker, stored
    in a float64 vector, plus the one-hot argmax;
  * `get_eer_auc`  (`evaluation.py:47-52`, Q19);
  * `Siamese.l2_dist` / `Siamese.forward`  (`siamese.py:10-30`, Q20).

`sklearn.metrics.pairwise.cosine_similarity` is third-party (unversioned in the
reference; scikit-learn 1.7.2 in this image): its published algorithm is
restated here -- L2-normalise each row (a zero norm is replaced by 1), then a
dot product, all in the input dtype.  `roc_curve`, `roc_auc_score`, `brentq`
and `interp1d` are called as the reference calls them; they are importable on
the GPU box too.

Parity status: PINNED by `tests/golden/scoring.npz` (made by
`tools/make_golden.py` calling the reference's own `compute_Similarity`,
`get_eer_auc`, `Siamese.l2_dist` and -- since round 3 -- `Siamese.forward`:
the reference's forward only needs the `.cuda()` ATTRIBUTE (`siamese.py:16,21`),
so the generator makes `torch.Tensor.cuda` the identity and runs it on the CPU
as written; `contrastive_loss` below is held to its losses at <= 1e-6 relative
(`sf_*` arrays, four (LAMBDA, M) pairs).
"""
import numpy as np


def _unit_rows(m):
    norms = np.sqrt(np.einsum("ij,ij->i", m, m))
    norms = np.where(norms == 0, 1, norms).astype(m.dtype)
    return m / norms[:, None]


def cosine_matrix(test, enroll):
    """(Nt, D) x (Ns, D) -> (Nt, Ns) cosine scores in float32 -- what the
    reference obtains pair by pair at evaluation.py:76-77."""
    t = _unit_rows(np.asarray(test, dtype=np.float32))
    e = _unit_rows(np.asarray(enroll, dtype=np.float32))
    return t @ e.T


def compute_similarity(embedding, enroll):
    """One utterance against every enrolled speaker.  evaluation.py:73-84.
    Returns (similarity_vec float64 (Ns,), assigned_speaker_vec float64 (Ns,))."""
    sims = np.zeros(len(enroll))
    for j in range(len(enroll)):
        sims[j] = cosine_matrix(np.asarray(embedding).reshape(1, -1),
                                np.asarray(enroll[j]).reshape(1, -1))[0, 0]
    assigned = np.zeros(len(enroll))
    assigned[np.argmax(sims)] = 1
    return sims, assigned


def get_eer_auc(label, distance):
    """evaluation.py:47-52: ROC on (label, score), AUC, and the EER as the root
    of 1 - x - tpr(x) on a linear interpolant of the ROC."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_auc_score, roc_curve
    fpr, tpr, _ = roc_curve(label, distance, pos_label=1)
    auc = roc_auc_score(label, distance)
    eer = brentq(lambda x: 1. - x - interp1d(fpr, tpr)(x), 0., 1.)
    return eer, auc, fpr, tpr


def k_fold_eer_auc(label, scores, k=1):
    """Mean EER/AUC over k equal consecutive slices.  evaluation.py:11-33
    (the plotting at :16-42 is not part of the numbers)."""
    step = int(label.shape[0] / float(k))
    eers, aucs = [], []
    for s in range(k):
        eer, auc, _, _ = get_eer_auc(label[s * step:(s + 1) * step],
                                     scores[s * step:(s + 1) * step])
        eers.append(eer)
        aucs.append(auc)
    return float(np.mean(eers)), float(np.mean(aucs))


def l2_dist(o1, o2):
    """Row-wise Euclidean distance.  siamese.py:29-30."""
    d = np.asarray(o1, dtype=np.float32) - np.asarray(o2, dtype=np.float32)
    return np.sqrt(np.sum(d * d, axis=1, dtype=np.float32))


def contrastive_loss(y, o1, o2, param_norms, LAMBDA, M):
    """siamese.py:14-25: mean over the batch of
    y * 0.5 d^2 + (1 - y) * 0.5 max(0, M - d)^2 + LAMBDA * sum_p ||p||_2
    (the regulariser is added to every sample before the mean)."""
    y = np.asarray(y, dtype=np.float32)
    d = l2_dist(o1, o2)
    gen = 0.5 * d ** 2
    imp = 0.5 * np.maximum(np.float32(0.0), np.float32(M) - d) ** 2
    reg = np.float32(LAMBDA) * np.float32(np.sum(param_norms))
    return float((1.0 / y.shape[0]) * np.sum(y * gen + (1 - y) * imp + reg))
