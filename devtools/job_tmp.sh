set -e
for r in 1 2; do
for rows in 16 32 64; do
echo "sdev rows fixed at $rows"
MUSICA_TUNE_SDEV=0 MUSICA_SDEV_ROWS=$rows TIF_LINEAR=1 TIF_MAX=3 python devtools/two_in_flight.py | grep "flight 3\|flight 1" | tail -2
done
echo "autotuned"
TIF_LINEAR=1 TIF_MAX=3 python devtools/two_in_flight.py | grep "flight 3\|flight 1" | tail -2
done
