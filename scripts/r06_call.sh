timeout -k 10 1100 python3 -m pytest tests -q -x -m gpu 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
