# scratch: the command of the moment for one gpurun call (`gpurun -- 'bash tools/job.sh'`); overwritten at will
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
