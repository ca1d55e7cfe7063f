"""Poor man's pyflakes (none is installed here): names that a function of a module loads but that are bound nowhere -- not in the
function, its enclosing functions, the module, or builtins.  usage: undefined_names.py file.py ..."""
import ast, builtins, sys


def bound_names(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
    return out


def main():
    bad = 0
    for path in sys.argv[1:]:
        tree = ast.parse(open(path).read(), path)
        known = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        for n in ast.walk(tree):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known:
                print("%s:%d: undefined name %r" % (path, n.lineno, n.id))
                bad += 1
    sys.exit(1 if bad else 0)


main()
