mkdir -p gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ITERS=2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc3 -o p1 -- python scripts/stage_times.py > gpurun_out/pmc3/log1.txt 2>&1
python scripts/rocpd_pmc.py gpurun_out/pmc3/p1_results.db k_bpm_band
ITERS=2 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmc3 -o p2 -- python scripts/stage_times.py > gpurun_out/pmc3/log2.txt 2>&1
python scripts/rocpd_pmc.py gpurun_out/pmc3/p2_results.db k_bpm_band
python scripts/rocpd_stats.py gpurun_out/pmc3/p2_results.db | head -14 | cut -c1-64,105-190
