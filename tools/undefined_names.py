"""python tools/undefined_names.py [paths...] : names that are read but bound nowhere (module scope, enclosing functions, builtins) -- the
image has no pyflakes, and a round that deletes whole alternatives (round 5: every CPU / torch branch of the product) can leave a reference
behind in a branch no test reaches.  Scope rules simplified: a name bound ANYWHERE in a function counts as bound in it.  Also checked, for
this repository's own modules: `module.name` and `from module import name` exist, and calls of plain module-level functions fit their
definitions -- which covers the GPU tests and tools that no CPU run executes."""
import ast
import builtins
import os
import sys


def bound_names(node):
    """names bound directly in this scope (not in nested functions / classes / lambdas; comprehensions are folded into their scope)"""
    out = set()

    def visit(n, top=False):
        if not top and isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
            return
        if not top and isinstance(n, ast.Lambda):
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, (ast.MatchAs, ast.MatchStar)) and getattr(n, "name", None):
            out.add(n.name)
        for c in ast.iter_child_nodes(n):
            visit(c)

    visit(node, top=True)
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    problems = []
    module_names = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__builtins__"}
    star = any(isinstance(n, ast.ImportFrom) and any(a.name == "*" for a in n.names) for n in ast.walk(tree))

    def walk(node, scopes):
        for c in ast.iter_child_nodes(node):
            if isinstance(c, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                for d in getattr(c, "decorator_list", []):
                    walk_expr(d, scopes)
                args = c.args
                for d in list(args.defaults) + [d for d in args.kw_defaults if d is not None]:
                    walk_expr(d, scopes)
                walk(c, scopes + [bound_names(c)])
            elif isinstance(c, ast.ClassDef):
                for d in c.decorator_list + c.bases + [k.value for k in c.keywords]:
                    walk_expr(d, scopes)
                # class bodies see their own names while executing, methods do not: approximate by checking the body against class + outer names
                walk(c, scopes + [bound_names(c)])
            else:
                walk_expr(c, scopes, recurse=False)
                walk(c, scopes)

    def walk_expr(n, scopes, recurse=True):
        nodes = ast.walk(n) if recurse else [n]
        for m in nodes:
            if isinstance(m, ast.Name) and isinstance(m.ctx, ast.Load):
                if not any(m.id in s for s in scopes) and not star:
                    problems.append((path, m.lineno, m.id))

    walk(tree, [module_names])
    return problems


# ---- attributes of this repository's own modules: `ops.join_if_returned`, `L._outer` ... must exist in the module they are read from
PACKAGES = ("neusky_amd", "oracle", "util_step", "bench", "__graft_entry__")


def module_file(dotted):
    base = os.path.join(*dotted.split("."))
    for root in ("", "tests", "tools"):  # the tests and tools put their own directory on sys.path (util_step, bench helpers)
        for cand in (os.path.join(root, base + ".py"), os.path.join(root, base, "__init__.py")):
            if os.path.isfile(cand):
                return cand
    return None


_MODULE_NAMES: dict = {}


def names_of_module(path):
    if path not in _MODULE_NAMES:
        tree = ast.parse(open(path).read(), path)
        names = bound_names(tree)
        if any(isinstance(n, ast.ImportFrom) and any(a.name == "*" for a in n.names) for n in ast.walk(tree)) or "__getattr__" in names:
            names = None  # star imports / module __getattr__: anything goes
        elif os.path.basename(path) == "__init__.py":  # submodules are attributes of a package
            d = os.path.dirname(path)
            names |= {f[:-3] for f in os.listdir(d) if f.endswith(".py")} | {f for f in os.listdir(d) if os.path.isdir(os.path.join(d, f))}
        _MODULE_NAMES[path] = names
    return _MODULE_NAMES[path]


def check_attributes(path):
    """aliases bound by an import of one of PACKAGES' modules (anywhere in the file; a name also assigned elsewhere is skipped)"""
    tree = ast.parse(open(path).read(), path)
    here = os.path.dirname(os.path.relpath(path)).replace(os.sep, ".")
    alias, assigned, missing = {}, set(), []
    for n in ast.walk(tree):
        if isinstance(n, ast.ImportFrom):
            if n.level:
                parts = here.split(".") if here else []
                parts = parts[:len(parts) - (n.level - 1)] if n.level > 1 else parts
                mod = ".".join(parts + ([n.module] if n.module else []))
            else:
                mod = n.module or ""
            if mod.split(".")[0] not in PACKAGES:
                continue
            src = module_file(mod)
            for a in n.names:
                f = module_file(mod + "." + a.name)
                if f:  # `from pkg import module [as x]`
                    alias[a.asname or a.name] = f
                elif src and a.name != "*":  # `from pkg.module import name`: the name must be bound there
                    names = names_of_module(src)
                    if names is not None and a.name not in names:
                        missing.append((path, n.lineno, f"from {mod} import {a.name} (no such name in {src})"))
        elif isinstance(n, ast.Import):
            for a in n.names:
                if a.name.split(".")[0] in PACKAGES and a.asname and module_file(a.name):
                    alias[a.asname] = module_file(a.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            assigned.add(n.id)
        elif isinstance(n, ast.arg):
            assigned.add(n.arg)
    problems = missing
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id in alias and n.value.id not in assigned:
            names = names_of_module(alias[n.value.id])
            if names is not None and n.attr not in names and not n.attr.startswith("__"):
                problems.append((path, n.lineno, f"{n.value.id}.{n.attr} (no such name in {alias[n.value.id]})"))
    return problems


# ---- call shapes: a call of a plain module-level `def` of this repository (through a module alias, a from-import, or inside its own module)
# must fit the definition: not too many positionals, no unknown keyword, no required parameter left out
_MODULE_DEFS: dict = {}


def defs_of_module(path):
    if path not in _MODULE_DEFS:
        tree = ast.parse(open(path).read(), path)
        defs, rebound = {}, set()
        for n in tree.body:
            if isinstance(n, ast.FunctionDef) and not n.decorator_list:
                if n.name in defs:
                    rebound.add(n.name)
                defs[n.name] = n.args
        for n in ast.walk(tree):  # a name assigned anywhere else (monkeypatching, conditional definitions) is not checked
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store) and n.id in defs:
                rebound.add(n.id)
        _MODULE_DEFS[path] = {k: v for k, v in defs.items() if k not in rebound}
    return _MODULE_DEFS[path]


def call_problem(call, a):
    if any(isinstance(x, ast.Starred) for x in call.args) or any(k.arg is None for k in call.keywords):
        return None  # *args / **kwargs at the call site: not decidable here
    pos = [x.arg for x in a.posonlyargs + a.args]
    kwonly = [x.arg for x in a.kwonlyargs]
    if len(call.args) > len(pos) and a.vararg is None:
        return f"{len(call.args)} positional arguments for {len(pos)}"
    given = set(pos[:len(call.args)])
    for k in call.keywords:
        if k.arg in given:
            return f"argument {k.arg!r} given twice"
        if k.arg not in pos and k.arg not in kwonly and a.kwarg is None:
            return f"no parameter {k.arg!r}"
        given.add(k.arg)
    required = pos[:len(pos) - len(a.defaults)] + [n for n, d in zip(kwonly, a.kw_defaults) if d is None]
    left = [r for r in required if r not in given]
    return f"missing {left}" if left else None


def check_calls(path):
    tree = ast.parse(open(path).read(), path)
    here = os.path.dirname(os.path.relpath(path)).replace(os.sep, ".")
    alias, direct, assigned = {}, {}, set()
    for n in ast.walk(tree):
        if isinstance(n, ast.ImportFrom):
            if n.level:
                parts = here.split(".") if here else []
                parts = parts[:len(parts) - (n.level - 1)] if n.level > 1 else parts
                mod = ".".join(parts + ([n.module] if n.module else []))
            else:
                mod = n.module or ""
            if mod.split(".")[0] not in PACKAGES:
                continue
            src = module_file(mod)
            for a in n.names:
                f = module_file(mod + "." + a.name)
                if f:
                    alias[a.asname or a.name] = f
                elif src and a.name in defs_of_module(src):
                    direct[a.asname or a.name] = (src, a.name)
        elif isinstance(n, ast.Import):
            for a in n.names:
                if a.name.split(".")[0] in PACKAGES and a.asname and module_file(a.name):
                    alias[a.asname] = module_file(a.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            assigned.add(n.id)
        elif isinstance(n, ast.arg):
            assigned.add(n.arg)
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n not in tree.body:
            assigned.add(n.name)  # a nested def shadows
    own = defs_of_module(path)
    problems = []
    for n in ast.walk(tree):
        if not isinstance(n, ast.Call):
            continue
        f, target = n.func, None
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in alias and f.value.id not in assigned:
            target = (alias[f.value.id], f.attr, f"{f.value.id}.{f.attr}")
        elif isinstance(f, ast.Name) and f.id in direct and f.id not in assigned:
            target = (*direct[f.id], f.id)
        elif isinstance(f, ast.Name) and f.id in own and f.id not in assigned:
            target = (path, f.id, f.id)
        if target and target[1] in defs_of_module(target[0]):
            why = call_problem(n, defs_of_module(target[0])[target[1]])
            if why:
                problems.append((path, n.lineno, f"call {target[2]}(...): {why} (defined in {target[0]})"))
    return problems


def main(argv):
    roots = argv or ["neusky_amd", "bench.py", "__graft_entry__.py", "oracle", "tools", "tests"]
    files = []
    for r in roots:
        if os.path.isdir(r):
            for d, _, fs in os.walk(r):
                files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
        elif r.endswith(".py"):
            files.append(r)
    problems = []
    for f in sorted(files):
        problems += check(f)
        problems += check_attributes(f)
        problems += check_calls(f)
    for p, line, name in problems:
        print(f"{p}:{line}: undefined name {name!r}")
    print(f"{len(files)} files, {len(problems)} undefined names")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
