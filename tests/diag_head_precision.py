"""Diagnostic (run by name on the GPU box, not collected): where the relation head's error vs the fp32 oracle comes
from.  Teacher-forced (oracle head on the HIP path's own temporal maps) and free-running, per intermediate feature,
plus how the top-100 identity reacts to the size of the score error (oracle scores + synthetic noise)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_e2e_gpu import build_model, rel_l2, synth_labels_scores   # noqa: E402

DEV = "cuda:0"


def main():
    from oracle.head_ref import head_forward
    from oracle.i3d_ref import i3d_forward
    from oracle.np_oracle import topk_triplets
    from trace_amd.synthetic import make_inputs
    for regime in ("A", "B"):
        m = build_model(regime)
        B, T, n = 1, 8, 10
        frames, blob, feat, rois = make_inputs(B, T, n, 224, seed=40 + B + T)
        labels, scores = synth_labels_scores(rois, m.cfg.NUM_CLASSES, 7)
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        i3d_sd = {k[len("Prd_RCNN."):]: v for k, v in sd.items() if k.startswith("Prd_RCNN.")}
        m.to(DEV)
        outs = m.forward_clips(frames.to(DEV), blob.to(DEV), feat.to(DEV), rois, topk=100, prd_k=2, det_labels=labels, det_scores=scores)
        dbg = m._debug
        tm_hip = dbg["tmaps"].float().permute(0, 1, 4, 2, 3).cpu()
        tm = i3d_forward(i3d_sd, frames[:1].permute(0, 2, 1, 3, 4).contiguous())[0].permute(1, 0, 2, 3).contiguous()
        for tag, maps in (("teacher-forced", tm_hip[0]), ("free-running", tm)):
            ref = head_forward(sd, blob[:1], feat[:1], maps, rois[0], det_labels_in=labels[0], det_scores_in=scores[0])
            P = len(ref["sbj_inds"])
            names = [("det_feat", dbg["det_feat"], ref["det_feat"]), ("obj_agg", dbg["obj_agg"], ref["obj_agg"]),
                     ("reg_agg", dbg["reg_agg"], ref["reg_agg"]),
                     ("pf", dbg["pf"][dbg["node_obj"]], ref["pooled_features"][:n]),
                     ("obj_feature", dbg["obj_feature"], ref["obj_feature"]), ("q_s", dbg["q_s"], ref["q_s"]),
                     ("spatio_infer", dbg["spatio_infer"], torch.nn.functional.leaky_relu(ref["spatio_infer"], 0.1)),
                     ("logits", dbg["logits"], ref["logits"])]
            line = " ".join("%s %.4f" % (k, rel_l2(a, b)) for k, a, b in names)
            o = outs[0]
            rs = ref["det_scores"]
            full = ref["prd_scores"].numpy() * (rs[ref["sbj_inds"]] * rs[ref["obj_inds"]])[:, None]
            sc, pair, prd = topk_triplets(ref["prd_scores"].numpy(), rs[ref["sbj_inds"]], rs[ref["obj_inds"]], 2, 100)
            want = set(zip(pair.tolist(), prd.tolist()))
            got = set(zip(o["top_pair"].cpu().tolist(), o["top_prd"].cpu().tolist()))
            lerr = (dbg["logits"].float().cpu() - ref["logits"])
            print("regime %s %-14s %s | logits abs err max %.4f rms %.4f scale %.3f | identity %.3f" % (
                regime, tag, line, float(lerr.abs().max()), float(lerr.pow(2).mean().sqrt()), float(ref["logits"].std()),
                len(want & got) / 100.0))
            srt = np.sort(full.reshape(-1))[::-1]
            print("   scores around the cut (ranks 96..105):", " ".join("%.5f" % v for v in srt[95:105]))
            # identity as a function of the size of a relative score error
            rng = np.random.default_rng(0)
            for e in (1e-3, 3e-3, 1e-2, 2e-2):
                ids = []
                for _ in range(50):
                    noisy = ref["prd_scores"].numpy() * (1 + e * rng.standard_normal(ref["prd_scores"].shape)).astype(np.float32)
                    _, p2, k2 = topk_triplets(noisy, rs[ref["sbj_inds"]], rs[ref["obj_inds"]], 2, 100)
                    ids.append(len(want & set(zip(p2.tolist(), k2.tolist()))))
                print("   relative score noise %.0e -> mean identity %.2f / 100" % (e, np.mean(ids)))


if __name__ == "__main__":
    main()
