import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
for mode in ("s16", "h3", "s16", "h3"):
    os.environ["SCANERF_DECODER_FWD"] = mode
    r = bench.decoder_op_leg(dev, 65536 * 128)["decoder_op"]
    print(mode, round(r["forward_ms"], 3), round(r["backward_ms"], 3))
