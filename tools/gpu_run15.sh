#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad or elbo_value or multiclass" 2>&1 | tail -15
