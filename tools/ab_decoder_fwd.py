"""A/B of the stand-alone decoder forward: 32-sample tiles (k_decoder_fwd_h3, default) against 16-sample tiles
(k_decoder_fwd_s16, SCANERF_DECODER_FWD=s16), same box, same inputs; outputs compared with each other.
Usage: python tools/ab_decoder_fwd.py [N]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scanerf_amd  # noqa: F401
from scanerf_amd import decoder_op

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536 * 128
dev = torch.device("cuda:0")
torch.manual_seed(3)
from scanerf_amd import network
x = torch.cat([0.3 * torch.randn(N, 32, device=dev), torch.randn(N, 3, device=dev)], -1)
blob = network.xavier_blob(1, dev, bias_scale=0.05)
wf = network.weight_feature(20000, dev)
res = {}
outs = {}
for mode in ("h3", "s16"):
    os.environ["SCANERF_DECODER_FWD"] = mode
    with torch.no_grad():
        for _ in range(3):
            o = decoder_op.decoder_apply(x, blob, wf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            o = decoder_op.decoder_apply(x, blob, wf)
        e1.record()
        torch.cuda.synchronize()
    res[mode + "_ms"] = e0.elapsed_time(e1) / 10
    outs[mode] = [t.clone() for t in o]
res["max_abs_diff"] = max(float((a - b).abs().max()) for a, b in zip(outs["h3"], outs["s16"]))
print(json.dumps(res))
