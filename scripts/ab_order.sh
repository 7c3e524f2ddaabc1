#!/bin/bash
# Same-box A/B of the cost-ordered raster workgroups (debug flags: 0 = product, 0x10000000 = off, 0x40000000 = order made
# anew with every frame, 0x20000000 = off with frames in flight). Single-stream columns are the ones that move.
mkdir -p gpurun_out/order
python scripts/ab_flags.py "${1:-dense scene surfel 5m_dense 5m_scene}" "${2:-0,0x10000000,0x40000000,0x20000000}" 2 > gpurun_out/order/ab_order.txt 2>&1
tail -45 gpurun_out/order/ab_order.txt
