#!/bin/bash
# round 4: counters of the occupancy variants of the forward kernel, old vs new on ONE box
# (VERDICT r03 item 2): SQ wait / active / wave cycles, LDS, instruction counts, GRBM_GUI_ACTIVE
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r04_occupancy_counters.txt
{
echo "# tools/pmc_occupancy.sh $(date -u +%FT%TZ): rocprofv3 --pmc (tools/wfft/pmc.sh), 24 GB of input per launch, 2 launches each"
echo "## A: library kernel, plan R0=20 R=1 (10000 frames), 1 workgroup / CU, 252 VGPR, 2 waves / SIMD"
WF_R0=20 WF_R=1 bash tools/wfft/pmc.sh A_r20 "" time 150000 10000 2 0
echo "## B: plan R0=10 R=2 (10000 frames as 2 x 5120, four passes), library LONG kernel, 1 workgroup / CU, 218 VGPR"
WF_R0=10 WF_R=2 bash tools/wfft/pmc.sh B_r10x2 "" time 150000 10000 2 0
echo "## C: the same compiled for 3 waves / SIMD (168 VGPR, 176 B scratch; LDS still admits 2 workgroups / CU, registers 1.5)"
WF_R0=10 WF_R=2 bash tools/wfft/pmc.sh C_r10x2_mw3 mw3 time 150000 10000 2 0
echo "## D: the same compiled for 4 waves / SIMD (128 VGPR, 360 B scratch), 2 workgroups / CU"
WF_R0=10 WF_R=2 bash tools/wfft/pmc.sh D_r10x2_mw4 mw4 time 150000 10000 2 0
echo "## E: plan R0=8 (4096 frames), 126 VGPR: 2 workgroups / CU as the library runs it"
WF_R0=8 WF_R=1 bash tools/wfft/pmc.sh E_r8_2wg "" time 360000 4096 2 0
echo "## F: the SAME binary held to 1 workgroup / CU (WF_PERCU=1)"
WF_PERCU=1 WF_R0=8 WF_R=1 bash tools/wfft/pmc.sh F_r8_1wg "" time 360000 4096 2 0
} > $O 2>&1
cat $O
