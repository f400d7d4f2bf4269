"""CPU restatement (torch fp32, same ATen ops in the reference's order) of the graph modules of the reference's fork
model lanercnn.py -- TEST INFRASTRUCTURE ONLY: imported by tests/ as the checker, never by the product.
Pinned by tests/golden/lanercnn_b3.npz, which tests/golden/make_golden.py produces by running the reference's own
modules (lanercnn.py imported with the shims of SURVEY.md Appendix A plus a torchvision stub)."""
import torch
import torch.nn.functional as F

import numpy as np

from .lanegcn_oracle import _gn, lane_conv, linear_block


def seeded_state(shapes, seed):
    """Deterministic weights for a list of (name, shape) in state_dict order (both sides regenerate them: the
    fixture stores no weights): GroupNorm gains around 1, other vectors small, matrices ~ N(0, 1/fan_in)."""
    r = np.random.RandomState(seed)
    sd = {}
    for k, shape in shapes:
        shape = tuple(shape)
        if k.endswith("norm.weight") or k.endswith("bn.weight"):
            a = 1.0 + 0.2 * r.randn(*shape)
        elif len(shape) == 1:
            a = 0.2 * r.randn(*shape)
        else:
            a = r.randn(*shape) / np.sqrt(shape[1])
        sd[k] = torch.from_numpy(a.astype(np.float32))
    return sd


def lane_input(graph, sd, prefix="input"):
    """LaneInput.forward, lanercnn.py:309-351."""
    map_feats = torch.cat(graph["feats"], 0)
    agt_feats = torch.cat(graph["agent_feat"], 0)
    out = F.linear(map_feats, sd[prefix + ".map_fc.weight"])
    out.index_add_(0, graph["a2m"]["v"], F.linear(agt_feats[graph["a2m"]["u"]], sd[prefix + ".agt_fc.weight"]))
    return F.relu(_gn(out, sd, prefix + ".bn"))


def lane_roi(feat, graph, sd, prefix="roi", num_scales=6):
    """LaneRoI.forward, lanercnn.py:384-430: input Linear(+GN+ReLU), then the 4-layer fuse loop (identical to MapNet's)."""
    feat = linear_block(feat, sd, prefix + ".input")
    return lane_conv(feat, graph, sd, prefix + ".fuse", num_scales)


def global_graph_net(feat, graph, sd, prefix="ggn", num_scales=6):
    """GlobalGraphNet.forward, lanercnn.py:547-600."""
    return lane_conv(feat, graph, sd, prefix + ".fuse", num_scales)


def lane_pooling(context_feat, context_graph, target_feat, target_graph, sd, prefix="pool", dist_th=6.0):
    """LanePooling.forward, lanercnn.py:462-514."""
    hi, wi, hc, wc = [], [], 0, 0
    for c, t in zip(context_graph["ctrs"], target_graph["ctrs"]):
        dist = c.view(-1, 1, 2) - t.view(1, -1, 2)
        dist = torch.sqrt((dist ** 2).sum(2))
        idcs = torch.nonzero(dist <= dist_th, as_tuple=False)
        if len(idcs) == 0:
            continue
        hi.append(idcs[:, 0] + hc)
        wi.append(idcs[:, 1] + wc)
        hc += len(c)
        wc += len(t)
    hi, wi = torch.cat(hi, 0), torch.cat(wi, 0)
    c_pose, t_pose = torch.cat(context_graph["pose"], 0), torch.cat(target_graph["pose"], 0)
    d = F.relu(F.linear(c_pose[hi] - t_pose[wi], sd[prefix + ".relpose.0.weight"], sd[prefix + ".relpose.0.bias"]))
    ctx = torch.cat([context_feat[hi], d], -1)
    ctx = F.linear(linear_block(ctx, sd, prefix + ".ctx.0"), sd[prefix + ".ctx.1.weight"])
    identity = target_feat
    out = F.linear(target_feat, sd[prefix + ".input.weight"])
    out.index_add_(0, wi, ctx)
    out = F.relu(_gn(out, sd, prefix + ".norm"))
    out = linear_block(linear_block(out, sd, prefix + ".mlp.0"), sd, prefix + ".mlp.1", act=False)
    return F.relu(out + identity), hi, wi
