bash tools/measure_all.sh r04m 2>&1 | tail -5
