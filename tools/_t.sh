#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/experiments/step_times.py
