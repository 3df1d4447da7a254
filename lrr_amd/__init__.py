"""lrr_amd — MI355X-native robust-ICP estimation inner loop of msofka/LRR.

Only what the hot path needs: ``csrc/`` (HIP kernels + C-ABI, host C++ flat view),
``estimation.py`` (ctypes view of the C++ drop-in classes), ``synth.py`` (seeded synthetic
feature clouds), ``multi_gpu.py`` (independent query locations sharded over ranks).
"""
