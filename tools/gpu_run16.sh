#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-250
