R=$PWD; rm -rf /tmp/tr2; mkdir -p /tmp/tr2 && cd /tmp/tr2
export PYTHONPATH=$R
python -m swift_amd.train experiment=era5-swinv2-1.4-scm data=era5-synthetic-1.4 data.batch_size=8 trainer.total_kimg=0.064 trainer.kimg_per_tick=0.032 trainer.val_ticks=null trainer.checkpoint_ticks=1 trainer.lr_rampup_kimg=0 > run1.log 2>&1
RID=$(ls results/era5-swinv2-1.4-scm | head -1); echo "run id $RID"
HYDRA_RUN_ID=001 python -m swift_amd.train experiment=era5-swinv2-1.4-scm data=era5-synthetic-1.4 resume=$RID finetune=multistep "finetune.finetune.intervals=[{steps: 4, kimg: 1}]" data.batch_size=8 trainer.total_kimg=0.264 trainer.kimg_per_tick=0.04 trainer.val_ticks=null trainer.checkpoint_ticks=null > run2.log 2>&1
grep -E "train/dt/kimg|Error|Traceback" run2.log | cut -c1-330 | tail -7
