#!/bin/bash
export TMPDIR=/tmp
timeout 400 python tools/bench_vit_order.py 2>&1 | tail -7
