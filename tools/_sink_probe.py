"""Scratch: where does the device sink's finalize spend its time on the shells cloud?"""
import sys, time, json
sys.path.insert(0, '.')
import torch
import mlsgpu_amd as m
from mlsgpu_amd import synth
import bench
dev = 0
args = bench.parse_args(["--dist", "shells"]) if hasattr(bench, "parse_args") else None
