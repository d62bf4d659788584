R=$PWD; rm -rf /tmp/tr4; mkdir -p /tmp/tr4 && cd /tmp/tr4
export PYTHONPATH=$R
nproc
for w in 4 16; do
HYDRA_RUN_ID=00$w python -m swift_amd.train experiment=era5-swinv2-1.4-trigflow data=era5-synthetic-1.4 data.batch_size=8 data.data_workers=$w trainer.total_kimg=0.8 trainer.kimg_per_tick=0.16 trainer.val_ticks=null trainer.checkpoint_ticks=null 2>&1 | grep -E "train/dt/kimg|Error|Traceback" | cut -c100-260 | tail -2
done
