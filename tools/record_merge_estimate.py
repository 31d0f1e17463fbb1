"""How many scatter records would merging equal cells of consecutive samples save?  (VERDICT r1 item 4 i.)
For the bench's configs[1] rays: per level, the number of (sample, level) cells of a 16-sample tile that equal the previous
sample's cell (same ray, same tile) -- those could ride in the previous record at the price of one record per x-entry
(two per (y,z) pair) instead of one per pair, because the x-weight differs from sample to sample."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd  # noqa
from scanerf_amd.tile_model import TileModel
dev = "cuda:0"
B, S = int(os.environ.get("B", 16384)), 128
torch.manual_seed(0)
m = TileModel([-4, -4, -4], [8, 8, 8], dev, log2_T=19)
o = torch.rand(B, 3, device=dev) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
z, _ = m.sample(o, d, S)
p = ((o[:, None] + z[..., None] * d[:, None]) - m._min_dev) / m._size_dev  # [B,S,3] in [0,1] of the 2x box
res = m.resolution.cpu().tolist()
tot_plain = tot_merged = 0.0
print("level  res   same-cell fraction   records now   records merged (2 per run)")
for l, r in enumerate(res):
    cell = torch.floor(p * torch.tensor(r, device=dev).float()).long()       # [B,S,3]
    same = (cell[:, 1:] == cell[:, :-1]).all(-1)                               # sample s equals s-1
    same = torch.cat([torch.zeros(B, 1, dtype=torch.bool, device=dev), same], 1)
    same[:, ::16] = False                                                      # a run does not cross a 16-sample tile
    frac = float(same.float().mean())
    plain = 4.0                                   # records per (sample, level)
    merged = 8.0 * (1 - frac)                     # runs x 4 pairs x 2 entries
    best = min(plain, merged)
    tot_plain += plain
    tot_merged += best
    print(f"{l:5d} {r[0]:5d}   {frac:8.3f}            {plain:4.1f}          {merged:5.2f}{'  <- pays' if merged < plain else ''}")
print(f"records per sample: {tot_plain:.1f} -> {tot_merged:.1f}  ({100 * (1 - tot_merged / tot_plain):.1f} % fewer)")
