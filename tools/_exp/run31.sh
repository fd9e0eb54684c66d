python tools/sweep.py 2 0 3 '{"a":{},"p":{"profile":1},"b":{}}' 1 2>&1 | grep -v "^    \(wave\|hwave\)"
python tools/sweep.py 2 16384 3 '{"f":{},"g":{}}' 1 2>&1 | grep -v "^   "
python tools/sweep.py 3 0 0 '{"a":{},"p":{"profile":1},"b":{}}' 1 2>&1 | grep -v "^    \(wave\|hwave\)"
python tools/sweep.py 4 0 0 '{"a":{},"p":{"profile":1},"b":{}}' 1 2>&1 | grep -v "^    \(wave\|hwave\)"
python tools/sweep.py 5 0 0 '{"a":{},"b":{}}' 1 2>&1 | grep -v "^   "
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
