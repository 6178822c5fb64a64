for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python scripts/vio_only.py 2>/dev/null | tail -1; done
