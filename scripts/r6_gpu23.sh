#!/bin/bash
python -m pytest tests/test_model_gpu.py -x -q -k "midsize" 2>&1 | tail -12
