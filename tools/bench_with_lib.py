"""bench.py against another build of the library: bench_with_lib.py <path to .so> [bench.py arguments]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_tf_amd as y
y._abi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
import bench
bench.main()
