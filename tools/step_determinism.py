"""Distinct results of N launches of the forward (with x-stash and tile_T outputs, as the training step calls it) on the same
inputs after PRE training steps; must be 1."""
import hashlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanerf_amd
from scanerf_amd import network, render
from scanerf_amd.tile_model import TileModel, train_step_fused
DEV = "cuda:0"
torch.manual_seed(9)
B, S = int(os.environ.get("B", 4096)), int(os.environ.get("S", 64))
VALID = bool(int(os.environ.get("VALID", "0")))  # pass the ray_valid mask as the training step does
PLAN = bool(int(os.environ.get("PLAN", "0")))  # the forward that also counts the scatter plan (k_render_fwd_h3<.., COUNT>)
o = torch.rand(B, 3, device=DEV) * 8 - 4
d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1) * (0.5 + torch.rand(B, 1, device=DEV))
tgt = torch.rand(B, 3, device=DEV)
dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[os.environ.get("DT", "bf16")]
m = TileModel([-4, -4, -4], [8, 8, 8], DEV, log2_T=int(os.environ.get("LOG2T", 14)), seed=1, table_dtype=dt)
with torch.no_grad():
    m.features.mul_(float(os.environ.get("SCALE", 30.0)))
opt = torch.optim.Adam(m.decoder.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
for i in range(int(os.environ.get("PRE", 2))):
    train_step_fused(m, opt, o, d, tgt, S, 20000 + i)
z, dist = m.sample(o, d, S)
m.packed.pack(m.decoder.blob(), m.weight_feature(20000))
box = (m.min_bbox.tolist(), m.bbox_size.tolist(), render.FORE, False)
table = m.gather_table()
valid = render.ray_valid(z) if VALID else None
EVICT = bool(int(os.environ.get("EVICT", "0")))
EV = torch.rand(1 << 16, device=DEV)
IDLE_US = int(os.environ.get("IDLE_US", "-1"))
REPACK = bool(int(os.environ.get("REPACK", "0")))  # re-pack the decoder image before every launch (as every step does)
ref_out = ref_xs = None
n_out = n_xs = 0
detail = 0
FLUSH = int(os.environ.get("FLUSH", "0"))  # MB written before every launch: the table is then cold in L2 / Infinity Cache
flush_buf = torch.empty(FLUSH << 20, dtype=torch.uint8, device=DEV) if FLUSH else None
for it in range(int(os.environ.get("N", 60))):
    if FLUSH:
        flush_buf.fill_(it & 255)
    tile_T = torch.empty(B, render.tile_T_columns(S), device=DEV)
    xs = torch.empty(B * S, 32, device=DEV)
    if EVICT:  # many different kernels before the launch: the forward's instructions are no longer in the instruction caches
        t = EV
        for f in (torch.sin, torch.cos, torch.exp, torch.erf, torch.tanh, torch.sigmoid, torch.sqrt, torch.abs, torch.floor, torch.ceil,
                  torch.log1p, torch.atan, torch.sinh, torch.cosh, torch.round, torch.trunc, torch.neg, torch.reciprocal, torch.square, torch.sign):
            t = f(t.abs() + 1.0)
        t = torch.cumsum(t, 0); t = torch.sort(t)[0]; t = t.half().float(); t = (t.view(256, -1) @ t.view(-1, 256)); t = torch.softmax(t, -1)
        t = t.to(torch.bfloat16).to(torch.float64).sum()
    if IDLE_US >= 0:  # let the GPU run dry before the launch (the training step does: compact_rays reads a count on the host)
        torch.cuda.synchronize()
        time.sleep(IDLE_US * 1e-6)
    if REPACK:
        m.packed.pack(m.decoder.blob(), m.weight_feature(20000))
    out = render.render_forward(o, d, z, dist, table, m.resolution, m.packed, *box, want_weights=False, tile_T=tile_T, xstash=xs, plan=PLAN,
                                ray_valid=valid)[0]
    if ref_out is None:
        ref_out, ref_xs = out.clone(), xs.clone()
        continue
    if not torch.equal(out, ref_out):
        n_out += 1
    if not torch.equal(xs, ref_xs):
        n_xs += 1
        if detail < int(os.environ.get("DETAIL", 3)):
            detail += 1
            dif = torch.nonzero(xs != ref_xs)
            rows = dif[:, 0].unique()
            print(f"  launch {it}: {dif.shape[0]} x-stash elements differ in {rows.numel()} samples; first (ray, sample): "
                  f"{[(int(r) // S, int(r) % S) for r in rows[:4]]}; positions {dif[dif[:, 0] == rows[0]][:, 1].tolist()}")
print(f"{os.environ.get('TAG', '')} table {dt}: launches whose out / x-stash differ from launch 0: {n_out} / {n_xs} of {int(os.environ.get('N', 60)) - 1}")
