#!/bin/bash
# a fixed set of generator / discriminator conv shapes through tools/bench_conv.py (A/B of kernel variants via env knobs)
set -e
b="python tools/bench_conv.py"
$b 16 256 256 32 16 3 1 1 0 30
$b 16 256 256 16 32 3 1 1 1 30
$b 16 256 256 16 8 3 1 1 0 30
$b 16 128 128 32 32 3 1 1 0 30
$b 16 128 128 32 32 3 1 1 1 30
$b 16 128 128 64 32 3 1 1 0 30
$b 16 128 128 32 64 3 1 1 1 30
$b 16 128 128 16 32 3 1 1 0 30
$b 16 64 64 64 64 3 1 1 0 30
$b 16 64 64 128 64 3 1 1 0 30
$b 16 64 64 64 128 3 1 1 1 30
$b 16 64 64 32 64 3 1 1 0 30
