# enumeration workgroups of 512 or 1024 threads by row width (PNGLOSS_HIP_ENUM_NT pins the size; PNGLOSS_HIP_ENUM_LDS=1 = the generous
# LDS bound of before, three 512-thread workgroups per CU instead of four)
cd ${GRAFT_REPO_ROOT:-.}
for W in ${ENUM_WIDTHS:-4096 3200 1920}; do
for NT in 512 1024; do
echo -n "W=$W NT=$NT: "; PNGLOSS_HIP_ENUM_NT=$NT python tests/tools/gpu_seg_time.py $W 1024 0 19 2 2 2>&1 | tail -1
done
done
