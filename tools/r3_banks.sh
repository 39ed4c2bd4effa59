set -e
export TMPDIR=/tmp
python3 tools/two_banks.py 32 1
python3 tools/two_banks.py 32 2
python3 tools/two_banks.py 32 2 lookahead=0
python3 tools/two_banks.py 32 4
python3 tools/two_banks.py 48 1
python3 tools/two_banks.py 48 2
python3 tools/two_banks.py 64 2
python3 tools/two_banks.py 64 1
