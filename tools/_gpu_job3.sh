bash tools/measure_all.sh r04n 2>&1 | tail -5
