import ast, sys, builtins, os
def check(path):
    src = open(path).read()
    tree = ast.parse(src)
    defined = set(dir(builtins))
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            defined.add(node.name)
            if not isinstance(node, ast.ClassDef):
                a = node.args
                for x in a.args + a.kwonlyargs + a.posonlyargs: defined.add(x.arg)
                if a.vararg: defined.add(a.vararg.arg)
                if a.kwarg: defined.add(a.kwarg.arg)
        elif isinstance(node, ast.Lambda):
            a = node.args
            for x in a.args + a.kwonlyargs + a.posonlyargs: defined.add(x.arg)
            if a.vararg: defined.add(a.vararg.arg)
            if a.kwarg: defined.add(a.kwarg.arg)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            defined.add(node.id)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            for n in node.names: defined.add((n.asname or n.name).split('.')[0])
        elif isinstance(node, ast.ExceptHandler) and node.name: defined.add(node.name)
        elif isinstance(node, (ast.Global, ast.Nonlocal)):
            for n in node.names: defined.add(n)
    defined |= {'__file__','__name__','__doc__'}
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in defined:
            print(f'{path}:{node.lineno}: undefined {node.id}')
for root in sys.argv[1:]:
    if os.path.isfile(root): check(root); continue
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith('.py'): check(os.path.join(d, f))
