# Same-box A/B of the t16s backward with libm's log1pf / expf and an IEEE division on the tile's chain (rounds 1-5) against
# v_exp / v_log / v_rcp (round 6).  Build first, here:
#   tools/build_variant.py libmhead render_bwd_t16="-DT16_LIBM_HEADS=1"; tools/build_variant.py plainhead render_bwd_t16=""
# then on the GPU box: bash tools/ab_heads.sh      (round 6: 5.25 / 5.24 -> 5.18 / 5.17 ms plan + backward; a second box 5.33 / 5.24 -> 5.26 / 5.23)
bash tools/ab_bwd.sh libmhead plainhead
