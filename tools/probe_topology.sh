#!/bin/bash
# What the box looks like to a placement decision: sockets, NUMA nodes, which node the GPU hangs off.
lscpu | grep -i "numa\|socket\|model name"
for d in /sys/class/drm/card*/device; do
    echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null) path=$(readlink -f $d)"
done
rocm-smi --showtopo 2>&1 | tail -20
python3 - <<'PY'
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
print("hipDeviceGetPCIBusId rc", hip.hipDeviceGetPCIBusId(buf, 64, 0), buf.value)
PY
cat /proc/meminfo | head -3
numactl --hardware 2>/dev/null | head
