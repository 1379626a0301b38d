set -e
cd "$(dirname "$0")"
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
$H -DPIPE -DLOADS=0 gemm_lab.hip -o /tmp/gl_c 2>/dev/null && /tmp/gl_c 1344
$H -DPIPE -DLOADS=0 -DNOLDS gemm_lab.hip -o /tmp/gl_e 2>/dev/null && /tmp/gl_e 1344
$H -DPIPE -DLOADS=0 -DNOLDS -DWM_=1 -DWN_=4 gemm_lab.hip -o /tmp/gl_f 2>/dev/null && /tmp/gl_f 1344
