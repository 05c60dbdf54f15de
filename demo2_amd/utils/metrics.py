"""Retrieval evaluator with the reference's call surface (reference: utils/metrics.py), computed on the GPU.

    evaluator = R1_mAP_eval(num_query, max_rank=50, feat_norm=cfg.TEST.FEAT_NORM)     # engine/processor.py:45,205
    evaluator.reset(); evaluator.update((feat, pid, camid, img_paths)); cmc, mAP, *_ = evaluator.compute()

    R1_mAP (:172-218) is the MSVR310 variant: update((feat, pid, camid, sceneid, img_path)), removal by scene id.

The numpy reference sorts every distance row on the CPU and walks it in Python; here features stay on the device:
normalise -> |q|^2 + |g|^2 - 2 q.g^T (fp32 GEMM) -> one workgroup per query sorts its row in LDS and reduces it to
(average precision, rank of the first hit) -- csrc/evalrank.hip.  Only nq doubles + nq ints come back to the host.
Plotting, t-SNE and re-ranking (off by default in the reference) are not provided.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def _dev_ids(x, device):
    return torch.as_tensor(np.asarray(x), dtype=torch.int64).to(device)


def euclidean_distance(qf: torch.Tensor, gf: torch.Tensor) -> torch.Tensor:
    """utils/metrics.py:395-401 -- squared euclidean distances (nq, ng), fp32, on the device (the reference returns numpy)."""
    qf, gf = qf.float().contiguous(), gf.float().contiguous()
    dist = ops.gemm(qf, gf, out_dtype=torch.float32, alpha=-2.0)          # fp32 generic kernel: -2 q.g^T
    return ops.add_outer_sum(dist, ops.row_sqnorm(qf), ops.row_sqnorm(gf))


def _rank_metrics(distmat, q_pids, g_pids, q_keys, g_keys, max_rank):
    dev = distmat.device
    num_q, num_g = distmat.shape
    if num_g < max_rank:
        max_rank = num_g
        print("Note: number of gallery samples is quite small, got {}".format(num_g))
    ap, first = ops.rank_eval(distmat, _dev_ids(q_pids, dev), _dev_ids(g_pids, dev), _dev_ids(q_keys, dev), _dev_ids(g_keys, dev))
    ap, first = ap.cpu().numpy(), first.cpu().numpy()
    valid = first >= 0
    num_valid_q = float(valid.sum())
    assert num_valid_q > 0, "Error: all query identities do not appear in gallery"
    ranks = np.arange(max_rank)[None, :]
    all_cmc = (first[valid][:, None] <= ranks).astype(np.float32).sum(0) / num_valid_q
    mAP = np.mean(ap[valid])
    return all_cmc, mAP


def eval_func(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50):
    """utils/metrics.py:110-169 (market1501 protocol).  distmat: (nq, ng) CUDA tensor (or numpy, uploaded)."""
    if not torch.is_tensor(distmat):
        distmat = torch.as_tensor(np.asarray(distmat), dtype=torch.float32).cuda()
    return _rank_metrics(distmat.float().contiguous(), q_pids, g_pids, q_camids, g_camids, max_rank)


def eval_func_msrv(distmat, q_pids, g_pids, q_camids, g_camids, q_sceneids, g_sceneids, max_rank=50):
    """utils/metrics.py:12-107 (MSVR310: same identity AND same scene removed; the rank-list text file is not written)."""
    if not torch.is_tensor(distmat):
        distmat = torch.as_tensor(np.asarray(distmat), dtype=torch.float32).cuda()
    return _rank_metrics(distmat.float().contiguous(), q_pids, g_pids, q_sceneids, g_sceneids, max_rank)


class R1_mAP_eval:
    def __init__(self, num_query, max_rank=50, feat_norm=True, reranking=False):
        if reranking:
            raise NotImplementedError("k-reciprocal re-ranking is outside the hot path (SURVEY.md section 2, row 8)")
        self.num_query, self.max_rank, self.feat_norm = num_query, max_rank, feat_norm
        self.reset()

    def reset(self):
        self.feats, self.pids, self.camids, self.img_paths = [], [], [], []

    def update(self, output):                      # called once for each batch
        feat, pid, camid, img_paths = output
        self.feats.append(feat.detach())           # stays on the device (the reference moves every batch to the CPU)
        self.pids.extend(np.asarray(pid))
        self.camids.extend(np.asarray(camid))
        self.img_paths.extend(img_paths if img_paths is not None else [])

    def _features(self):
        feats = torch.cat(self.feats, dim=0).float().contiguous()
        if self.feat_norm in (True, "yes"):
            print("The test feature is normalized")
            feats = ops.l2_normalize_rows(feats.clone())
        return feats

    def compute(self):                             # called after each epoch
        feats = self._features()
        qf, gf = feats[:self.num_query], feats[self.num_query:]
        q_pids, g_pids = np.asarray(self.pids[:self.num_query]), np.asarray(self.pids[self.num_query:])
        q_camids, g_camids = np.asarray(self.camids[:self.num_query]), np.asarray(self.camids[self.num_query:])
        print('=> Computing DistMat with euclidean_distance')
        distmat = euclidean_distance(qf, gf)
        cmc, mAP = eval_func(distmat, q_pids, g_pids, q_camids, g_camids, self.max_rank)
        return cmc, mAP, distmat, self.pids, self.camids, qf, gf


class R1_mAP(R1_mAP_eval):
    def __init__(self, num_query, max_rank=50, feat_norm='yes'):
        super().__init__(num_query, max_rank, feat_norm)

    def reset(self):
        super().reset()
        self.sceneids = []

    def update(self, output):
        feat, pid, camid, sceneid, img_path = output
        super().update((feat, pid, camid, img_path))
        self.sceneids.extend(np.asarray(sceneid))

    def compute(self):
        feats = self._features()
        qf, gf = feats[:self.num_query], feats[self.num_query:]
        nq = self.num_query
        distmat = euclidean_distance(qf, gf)
        cmc, mAP = eval_func_msrv(distmat, np.asarray(self.pids[:nq]), np.asarray(self.pids[nq:]), np.asarray(self.camids[:nq]),
                                  np.asarray(self.camids[nq:]), np.asarray(self.sceneids[:nq]), np.asarray(self.sceneids[nq:]),
                                  self.max_rank)
        return cmc, mAP, distmat, self.pids, self.camids, qf, gf
