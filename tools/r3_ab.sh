set -e
bash tools/r3_run.sh r3c
