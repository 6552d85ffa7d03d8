import sys; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sys, __graft_entry__ as g
from pathlib import Path
g.build_libslq(extra_flags=sys.argv[1].split(), out=Path(sys.argv[2]))
print("built", sys.argv[2])
