#!/bin/bash
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 --genco clock_probe.hip -o clock_probe.hsaco
