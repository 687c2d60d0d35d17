"""models.loss.MatchMotionLoss with the reference's interface (3D/models/loss.py:47-170, 273-345): the FORWARD value of the
training loss on the device -- focal loss of conf_matrix_pred and conf_matrix_gt_hat, match recall / precision, the L1 motion
term -- through libdiffreg_hip (csrc/train.hip).  No autograd graph is built (forward half): the
returned tensors are values, e.g. for a validation pass; the static evaluation metrics are those of diffreg_hip.metrics."""
import torch
import torch.nn as nn

from diffreg_hip import lib
from diffreg_hip.metrics import MatchMetrics, compute_nrfmr as _compute_nrfmr


class MatchMotionLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.focal_alpha = config["focal_alpha"]
        self.focal_gamma = config["focal_gamma"]
        self.pos_w = config["pos_weight"]
        self.neg_w = config["neg_weight"]
        self.mot_w = config["motion_weight"]
        self.mat_w = config["match_weight"]
        self.motion_loss_type = config["motion_loss_type"]
        self.match_type = config["match_type"]
        self.positioning_type = config["positioning_type"]
        self.registration_threshold = config["registration_threshold"]
        self.confidence_threshold_metric = config["confidence_threshold_metric"]
        self.inlier_thr = config["inlier_thr"]
        self.fmr_thr = config["fmr_thr"]
        self.mutual_nearest = config["mutual_nearest"]
        self.dataset = config["dataset"]

    @torch.no_grad()
    def forward(self, data):
        loss_info = {}
        loss = self.ge_coarse_loss(data, loss_info)
        loss_info.update({"loss": loss})
        return loss_info

    @staticmethod
    def _match_rows(match_gt, dev):
        """the per-pair [2, K] index tensors of data['coarse_matches'] -> one [K_total, 3] (b, i, j) list on the device"""
        rows = [torch.cat([torch.full((1, m.shape[1]), b, dtype=torch.int64, device=m.device), m.to(torch.int64)], 0).t() for b, m in enumerate(match_gt)]
        return torch.cat(rows, 0).to(dev) if rows else torch.zeros(0, 3, dtype=torch.int64, device=dev)

    def match_2_conf_matrix(self, matches_gt, matrix_pred):
        P, N, M = matrix_pred.shape
        return lib.match_matrix(self._match_rows(matches_gt, matrix_pred.device), P, N, M).to(matrix_pred.dtype)

    def compute_correspondence_loss(self, conf, conf_gt, weight=None):
        return lib.focal_loss(conf, conf_gt, weight, self.focal_alpha, self.focal_gamma, self.pos_w, self.neg_w, self.match_type)

    @staticmethod
    def compute_match_recall(conf_matrix_gt, match_pred):
        return lib.match_recall(conf_matrix_gt, match_pred)

    def ge_coarse_loss(self, data, loss_info, eval_metric=False):
        src_mask, tgt_mask = data["src_mask"], data["tgt_mask"]
        conf_matrix_pred = data["conf_matrix_pred"]
        match_gt = data["coarse_matches"]
        dev = conf_matrix_pred.device
        P, N, M = conf_matrix_pred.shape
        rows = self._match_rows(match_gt, dev)
        c_weight = (src_mask[:, :, None] * tgt_mask[:, None, :]).float()
        conf_matrix_gt = lib.match_matrix(rows, P, N, M)
        focal_coarse = self.compute_correspondence_loss(conf_matrix_pred, conf_matrix_gt, weight=c_weight)
        recall, precision = self.compute_match_recall(conf_matrix_gt, data["coarse_match_pred"])
        loss_info.update({"focal_coarse": focal_coarse, "recall_coarse": recall, "precision_coarse": precision})
        loss = self.mat_w * focal_coarse
        if self.mot_w > 0 and recall > 0.01:                  # (one host read of the recall, as in the reference's `if`)
            s_overlap_mask = torch.zeros(P, N, dtype=torch.bool, device=dev)
            s_overlap_mask[rows[:, 0], rows[:, 1]] = True
            flow = None
            if self.dataset == "4dmatch":
                flow = torch.zeros_like(data["s_pcd"])
                for i, cflow in enumerate(data["coarse_flow"]):
                    flow[i][: len(cflow)] = cflow
            l1_loss = lib.motion_l1(data["s_pcd"], data["R_s2t_pred"], data["t_s2t_pred"], data["batched_rot"], data["batched_trn"], s_overlap_mask,
                                    flow=flow)
            loss = loss + self.mot_w * l1_loss
        loss_matrix_gt_hat = self.compute_correspondence_loss(data["conf_matrix_gt_hat"], conf_matrix_gt, weight=c_weight)
        loss_info.update({"loss_matrix_gt_hat": loss_matrix_gt_hat})
        return loss + loss_matrix_gt_hat

    def forward_train(self, data):
        """ge_coarse_loss WITH a graph on the outputs of Pipeline.forward_train (either match type: the dual-softmax form of the focal term is its
        positive part, loss.py:301-307): focal(conf_matrix_pred) * match_weight
        [+ motion_weight * L1(R, t) when recall > 0.01] + focal(conf_matrix_gt_hat), each term a device kernel with its backward kernel."""
        from diffreg_hip import autograd as dag
        conf = data["conf_matrix_pred"]
        P, N, M = conf.shape
        dev = conf.device
        rows = self._match_rows(data["coarse_matches"], dev)
        conf_gt = lib.match_matrix(rows, P, N, M)
        hp = (self.focal_alpha, self.focal_gamma, self.pos_w, self.neg_w)
        focal_coarse = dag.focal_loss(conf, conf_gt, *hp, match_type=self.match_type)
        recall, precision = self.compute_match_recall(conf_gt, data["coarse_match_pred"])
        loss_info = {"focal_coarse": focal_coarse.detach(), "recall_coarse": recall, "precision_coarse": precision}
        loss = self.mat_w * focal_coarse
        if self.mot_w > 0 and recall > 0.01:
            ov = torch.zeros(P, N, dtype=torch.bool, device=dev)
            ov[rows[:, 0], rows[:, 1]] = True
            flow = None
            if self.dataset == "4dmatch":
                flow = torch.zeros_like(data["s_pcd"])
                for i, cflow in enumerate(data["coarse_flow"]):
                    flow[i][: len(cflow)] = cflow
            loss = loss + self.mot_w * dag.motion_l1(data["s_pcd"], data["R_s2t_pred"], data["t_s2t_pred"], data["batched_rot"], data["batched_trn"], ov, flow)
        hat_loss = dag.focal_loss(data["conf_matrix_gt_hat"], conf_gt, *hp, match_type=self.match_type)
        loss_info.update({"loss_matrix_gt_hat": hat_loss.detach(), "loss": loss + hat_loss})
        return loss_info

    # ---- the static evaluation metrics (loss.py:347-448; 3D/lib/tester.py:150-210) run in diffreg_hip.metrics
    compute_inlier_ratio = staticmethod(MatchMetrics.compute_inlier_ratio)
    ransac_regist_coarse = staticmethod(MatchMetrics.ransac_regist_coarse)
    compute_registration_recall = staticmethod(MatchMetrics.compute_registration_recall)
    compute_nrfmr = staticmethod(_compute_nrfmr)
