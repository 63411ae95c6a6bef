#!/bin/bash
# dev: phase stamps of two instrumented builds of the Sinkhorn backward
cd $(dirname $0)/..
python tools/sinkhorn_stamps.py skst0 2>&1 | grep -v amdgpu
python tools/sinkhorn_stamps.py skst 2>&1 | grep -v amdgpu
