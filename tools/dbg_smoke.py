import faulthandler, sys, os
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.smoke()
