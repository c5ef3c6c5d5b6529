cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $O
rm -rf /tmp/pmc1 /tmp/pmc2
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc1 -- python tools/pmc_ws_run.py < /dev/null > $O/pmc1.out 2> $O/pmc1.err; echo "pmc1 rc=$?"
python tools/parse_pmc_generic.py /tmp/pmc1 $O/pmc1.json igemm
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc2 -- python tools/pmc_ws_run.py < /dev/null > $O/pmc2.out 2> $O/pmc2.err; echo "pmc2 rc=$?"
python tools/parse_pmc_generic.py /tmp/pmc2 $O/pmc2.json igemm
tail -3 $O/pmc1.err $O/pmc2.err
