#!/bin/bash
cd /root/repo
python -m pytest tests/test_wire_protocol.py -m gpu -x -q -k general_mesh 2>&1 | grep -E "^E  |passed|failed" | head -14
