import faulthandler, sys, os
faulthandler.dump_traceback_later(int(os.environ.get('DBG_T', '60')), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py'] + sys.argv[1:]
import bench
bench.main()
