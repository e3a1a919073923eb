for v in "" "BEATAMD_GS_ML=0"; do echo "== $v"; env $v python tools/time_config4.py 120 multilinear 2>&1 | grep "^issue\|^no timers" | tail -2 | cut -c1-330; done
