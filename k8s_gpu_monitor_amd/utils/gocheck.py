"""Static consistency check of the Go bindings against the C headers, for hosts without a Go toolchain.

What a `go vet` / `go build` of a cgo package would catch first, done on the text:

* every ``C.name`` used in a Go file is a cgo builtin, or is declared by the headers
  the file's preamble includes (function, type, enum constant or object-like macro);
* a call ``C.fn(a, b)`` passes as many arguments as the prototype takes;
* a member read or written on a variable of a C struct type exists in that struct
  (cgo spells members that are Go keywords with a leading underscore);
* braces, brackets and parentheses balance outside strings and comments, the file
  starts with a package clause, every import is used and no well-known standard
  package is used without being imported;
* no two files of a package define the same top-level function or type;
* ``//export`` names are Go functions of the package, and the C files of the package
  only call functions that exist;
* every ``pkg.Name`` a sample takes from one of the binding packages is an exported
  top-level name of that package.

It is not a compiler: types of expressions are not inferred beyond declared
variables, so a wrong argument type passes.  The checks are the ones that caught
real drift between gpumon.h and the bindings.
"""
import os
import re
from collections import defaultdict

GO_KEYWORDS = {"type", "func", "range", "map", "chan", "var", "go", "select", "package", "import", "interface", "defer",
               "fallthrough", "goto", "switch", "case", "default", "struct", "const", "return", "if", "else", "for",
               "break", "continue"}
CGO_BUILTIN = {"char", "schar", "uchar", "short", "ushort", "int", "uint", "long", "ulong", "longlong", "ulonglong",
               "float", "double", "size_t", "ssize_t", "int8_t", "int16_t", "int32_t", "int64_t", "uint8_t", "uint16_t",
               "uint32_t", "uint64_t", "uintptr_t", "intptr_t", "GoString", "GoStringN", "GoBytes", "CString", "CBytes"}
LIBC = {"free": 1, "malloc": 1, "calloc": 2, "memset": 3, "memcpy": 3, "strlen": 1}
STDLIB = {"fmt", "os", "time", "strings", "strconv", "math", "sync", "atomic", "unsafe", "errors", "log", "sort", "bytes",
          "bufio", "io", "json", "http", "context", "flag", "regexp", "exec", "filepath", "template", "signal", "syscall",
          "runtime", "reflect", "net", "ioutil", "tabwriter", "mux"}


# ------------------------------------------------------------------ C side --
def strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def _struct_members(body):
    """{member: type-name or None} of the top level of a struct body (nested aggregates count as one member)."""
    members = {}
    i, depth, cur = 0, 0, ""
    while i < len(body):
        ch = body[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == ";" and depth == 0:
            decl = cur.strip()
            cur = ""
            if decl:
                nested = "{" in decl
                tail = decl[decl.rfind("}") + 1:] if nested else decl
                names = []
                for part in _split_top(tail):
                    part = re.sub(r"\[[^\]]*\]", "", part)
                    part = re.sub(r":\s*\d+\s*$", "", part)
                    m = re.search(r"([A-Za-z_]\w*)\s*$", part.strip())
                    if m:
                        names.append(m.group(1))
                ctype = None
                if not nested:
                    m = re.match(r"\s*(?:const\s+)?(?:struct\s+|enum\s+)?([A-Za-z_]\w*)", tail)
                    ctype = m.group(1) if m else None
                for n in names:
                    members[n] = ctype
        else:
            cur += ch
        i += 1
    return members


class CDecls:
    def __init__(self):
        self.functions = {}   # name -> arity (None: variadic)
        self.constants = set()
        self.types = set()
        self.structs = {}     # typedef name -> {member: type}
        self.includes = set()

    def known(self, name):
        return name in self.functions or name in self.constants or name in self.types

    def merge(self, other):
        self.functions.update(other.functions)
        self.constants |= other.constants
        self.types |= other.types
        self.structs.update(other.structs)
        self.includes |= other.includes
        return self


def parse_c(text):
    d = CDecls()
    text = strip_c_comments(text)
    text = re.sub(r"\\\n", " ", text)
    for m in re.finditer(r'^[ \t]*#\s*include\s*[<"]([^>"]+)[>"]', text, flags=re.M):
        d.includes.add(m.group(1))
    for m in re.finditer(r"^[ \t]*#\s*define\s+([A-Za-z_]\w*)(\(?)", text, flags=re.M):
        if m.group(2) == "(":
            line_end = text.find("\n", m.end())
            params = text[m.end():text.find(")", m.end())]
            d.functions[m.group(1)] = len([p for p in params.split(",") if p.strip()])
            del line_end
        else:
            d.constants.add(m.group(1))
    text = re.sub(r"^[ \t]*#[^\n]*", "", text, flags=re.M)
    text = re.sub(r'extern\s+"C"\s*\{', "", text)
    # enums
    for m in re.finditer(r"\benum\b[^{;]*\{([^}]*)\}", text):
        for item in m.group(1).split(","):
            name = item.split("=")[0].strip()
            if re.fullmatch(r"[A-Za-z_]\w*", name or ""):
                d.constants.add(name)
    # typedef struct/union { ... } name;
    for m in re.finditer(r"\btypedef\s+(struct|union)\s*(\w*)\s*\{", text):
        depth, i = 1, m.end()
        while i < len(text) and depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        tail = re.match(r"\s*([A-Za-z_]\w*)\s*;", text[i:])
        if tail:
            d.types.add(tail.group(1))
            d.structs[tail.group(1)] = _struct_members(text[m.end():i - 1])
    # typedef enum {...} name;   typedef T name;   typedef R (*name)(...);
    for m in re.finditer(r"\btypedef\s+enum\b[^{;]*\{[^}]*\}\s*([A-Za-z_]\w*)\s*;", text):
        d.types.add(m.group(1))
    for m in re.finditer(r"\btypedef\s+[^;{}()]*?\(\s*\*\s*([A-Za-z_]\w*)\s*\)\s*\([^;]*\)\s*;", text):
        d.types.add(m.group(1))
    for m in re.finditer(r"\btypedef\s+[^;{}()]+?\b([A-Za-z_]\w*)\s*;", text):
        d.types.add(m.group(1))
    # functions at brace depth 0: "type name(args)" followed by ; or {
    flat, depth, out = text, 0, []
    for ch in flat:
        if ch == "{":
            depth += 1
            out.append("{" if depth == 1 else " ")
        elif ch == "}":
            depth -= 1
            out.append("}" if depth == 0 else " ")
        else:
            out.append(ch if depth == 0 else (" " if ch != "\n" else "\n"))
    top = "".join(out)
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b([A-Za-z_]\w*)\s*\(([^;{}]*?)\)\s*(;|\{)", top):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if not ret or ret.split()[0] == "typedef" or name in ("if", "while", "for", "switch", "return", "sizeof") or "(" in ret:
            continue
        if "(*" in args and args.count("(") != args.count(")"):
            continue
        parts = [p for p in _split_top(args) if p.strip()]
        if args in ("", "void"):
            d.functions[name] = 0
        elif parts and parts[-1].strip() == "...":
            d.functions[name] = None
        else:
            d.functions[name] = len(parts)
    return d


# ----------------------------------------------------------------- Go side --
def strip_go(text):
    """Go source with comments and the contents of string / rune literals blanked (same length, newlines kept)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        two = text[i:i + 2]
        if two == "//":
            j = text.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif two == "/*":
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append(re.sub(r"[^\n]", " ", text[i:j]))
            i = j
        elif c == "`":
            j = text.find("`", i + 1)
            j = n if j < 0 else j + 1
            out.append("`" + re.sub(r"[^\n]", " ", text[i + 1:j - 1]) + "`")
            i = j
        elif c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c and text[j] != "\n":
                j += 2 if text[j] == "\\" else 1
            j = min(n, j + 1)
            out.append(c + " " * (j - i - 2) + c if j - i >= 2 else c)
            i = j
        else:
            out.append(c)
            i += 1
    return "".join(out)


def cgo_preamble(text):
    m = re.search(r"/\*(.*?)\*/\s*import\s+\"C\"", text, flags=re.S)
    if m:
        return m.group(1)
    m = re.search(r"((?:^//[^\n]*\n)+)import\s+\"C\"", text, flags=re.M)
    return re.sub(r"^//", "", m.group(1), flags=re.M) if m else None


def _matching(code, i, open_ch="(", close_ch=")"):
    depth = 0
    for j in range(i, len(code)):
        if code[j] == open_ch:
            depth += 1
        elif code[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j
    return -1


def _count_args(arg_text):
    if not arg_text.strip():
        return 0
    return len(_split_top(arg_text))


def _imports(code, raw):
    """[(local name, path)] of a file; code = stripped text, raw = original (for the paths)."""
    out = []
    for m in re.finditer(r"^import\s*(\(|[\w\.]*\s*\")", code, flags=re.M):
        if m.group(1) == "(":
            end = _matching(code, m.end() - 1)
            block = raw[m.end():end]
        else:
            block = raw[m.start() + len("import"):raw.find("\n", m.start())]
        for im in re.finditer(r"(?:([\w\.]+)\s+)?\"([^\"]+)\"", block):
            alias, path = im.group(1), im.group(2)
            name = alias or path.rstrip("/").split("/")[-1]
            out.append((name, path))
    return out


def _functions(code):
    """[(name, receiver or None, header text, body text, body offset)] of the top-level functions."""
    out = []
    for m in re.finditer(r"^func\s*(\([^)]*\))?\s*([A-Za-z_]\w*)?\s*\(", code, flags=re.M):
        close = _matching(code, m.end() - 1)
        brace = code.find("{", close)
        nl = code.find("\n", close)
        if brace < 0 or (0 <= nl < brace and "{" not in code[close:nl]):
            continue
        end = _matching(code, brace, "{", "}")
        out.append((m.group(2), m.group(1), code[m.start():brace], code[brace:end + 1], brace))
    return out


_DECL_PATTERNS = [
    (re.compile(r"\bvar\s+([A-Za-z_]\w*)\s+C\.(\w+)"), "val"),
    (re.compile(r"\bvar\s+([A-Za-z_]\w*)\s+\[[^\]]*\]C\.(\w+)"), "seq"),
    (re.compile(r"\b([A-Za-z_]\w*)\s*:?=\s*&?C\.(\w+)\s*\{"), "val"),
    (re.compile(r"\b([A-Za-z_]\w*)\s*:?=\s*make\(\s*\[\]C\.(\w+)"), "seq"),
    (re.compile(r"\b([A-Za-z_]\w*)\s*:?=\s*\(\*C\.(\w+)\)\("), "val"),
    (re.compile(r"\b([A-Za-z_]\w*)\s*:?=\s*\(\*\[[^\]]*\]C\.(\w+)\)\("), "seq"),
    (re.compile(r"[(,]\s*([A-Za-z_]\w*)\s+\*?C\.(\w+)\s*[,)]"), "val"),
    (re.compile(r"[(,]\s*([A-Za-z_]\w*)\s+\[\]C\.(\w+)\s*[,)]"), "seq"),
]


def _var_types(header, body, decls):
    """{variable: (struct typedef name, 'val' | 'seq')} declared in one function."""
    vt = {}
    scope = header + "\n" + body
    for pat, kind in _DECL_PATTERNS:
        for m in pat.finditer(scope):
            if m.group(2) in decls.structs:
                vt[m.group(1)] = (m.group(2), kind)
    changed = True
    while changed:
        changed = False
        # p := &arr[i]   /   for _, p := range arr   /   p := &x.member[i]   /   p := &x.member
        for m in re.finditer(r"\b([A-Za-z_]\w*)\s*:?=\s*&?([A-Za-z_]\w*)\[[^\]]*\]", scope):
            src = vt.get(m.group(2))
            if src and src[1] == "seq" and m.group(1) not in vt:
                vt[m.group(1)] = (src[0], "val")
                changed = True
        for m in re.finditer(r"\bfor\s+\w+\s*,\s*([A-Za-z_]\w*)\s*:=\s*range\s+([A-Za-z_]\w*)\b", scope):
            src = vt.get(m.group(2))
            if src and src[1] == "seq" and m.group(1) not in vt:
                vt[m.group(1)] = (src[0], "val")
                changed = True
        for m in re.finditer(r"\b([A-Za-z_]\w*)\s*:?=\s*&?([A-Za-z_]\w*)\.(\w+)(\[[^\]]*\])?\s*(?:\n|$|;)", scope):
            src = vt.get(m.group(2))
            if not src or m.group(1) in vt:
                continue
            mtype = decls.structs[src[0]].get(m.group(3).lstrip("_") if m.group(3).lstrip("_") in GO_KEYWORDS else m.group(3))
            if mtype in decls.structs:
                vt[m.group(1)] = (mtype, "val")
                changed = True
    return vt


def _member_name(go_name):
    return go_name[1:] if go_name.startswith("_") and go_name[1:] in GO_KEYWORDS else go_name


def check_go_file(path, decls, problems, stats=None):
    stats = stats if stats is not None else defaultdict(int)
    raw = open(path).read()
    code = strip_go(raw)
    rel = path

    def problem(pos, msg):
        problems.append(f"{rel}:{code.count(chr(10), 0, pos) + 1}: {msg}")

    # ---- syntax sanity
    first = re.search(r"\S+", code)
    if not first or first.group(0) != "package":
        problem(0, "file does not start with a package clause")
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for i, ch in enumerate(code):
        if ch in "([{":
            stack.append((ch, i))
        elif ch in ")]}":
            if not stack or stack[-1][0] != pairs[ch]:
                problem(i, f"unbalanced '{ch}'")
                stack = []
                break
            stack.pop()
    if stack:
        problem(stack[-1][1], f"unclosed '{stack[-1][0]}'")
    imports = _imports(code, raw)
    body_wo_imports = re.sub(r"^import\s*\([^)]*\)", lambda m: re.sub(r"[^\n]", " ", m.group(0)), code, flags=re.M)
    body_wo_imports = re.sub(r"^import\s+[^\n]*", "", body_wo_imports, flags=re.M)
    declared_locals = set(re.findall(r"\b([a-z]\w*)\s*:=", code)) | set(re.findall(r"\bvar\s+([a-z]\w*)", code)) | \
        set(re.findall(r"[(,]\s*([a-z]\w*)\s+[\*\[\]\w\.]+\s*[,)]", code))
    for name, ipath in imports:
        if name in ("C", "_", "."):
            continue
        if not re.search(r"\b%s\." % re.escape(name), body_wo_imports):
            problem(0, f'import "{ipath}" is not used')
    imported = {n for n, _ in imports}
    for m in re.finditer(r"(?<![\w\.])([a-z]\w*)\.[A-Z]\w*", body_wo_imports):
        pkg = m.group(1)
        if pkg in STDLIB and pkg not in imported and pkg not in declared_locals:
            problem(m.start(), f'package "{pkg}" is used but not imported')

    # ---- cgo
    uses_c = any(n == "C" for n, _ in imports)
    if not uses_c:
        if re.search(r"(?<![\w\.])C\.[A-Za-z_]", code):
            problem(0, 'C.* is used without import "C"')
        return
    for m in re.finditer(r"(?<![\w\.])C\.([A-Za-z_]\w*)", code):
        name = m.group(1)
        after = code[m.end():m.end() + 1]
        stats["c_references"] += 1
        if name in CGO_BUILTIN or re.match(r"(struct|union|enum|sizeof)_", name):
            continue
        if after == "(":
            close = _matching(code, m.end())
            nargs = _count_args(code[m.end() + 1:close])
            if name in decls.functions or name in LIBC:
                stats["calls_with_arity_checked"] += 1
                want = decls.functions.get(name, LIBC.get(name))
                if want is not None and nargs != want:
                    problem(m.start(), f"C.{name} takes {want} argument(s), called with {nargs}")
            elif name in decls.types:
                if nargs != 1:
                    problem(m.start(), f"conversion C.{name}(...) with {nargs} arguments")
            else:
                problem(m.start(), f"C.{name} is not declared by the headers")
        elif not decls.known(name) and name not in LIBC:
            problem(m.start(), f"C.{name} is not declared by the headers")

    # ---- struct members
    for fname, _recv, header, body, off in _functions(code):
        vt = _var_types(header, body, decls)
        for var, (stype, kind) in vt.items():
            pat = r"(?<![\w\.])%s%s\.([A-Za-z_]\w*)" % (re.escape(var), r"\[[^\]]*\]" if kind == "seq" else "")
            for m in re.finditer(pat, body):
                member = _member_name(m.group(1))
                stats["member_accesses_checked"] += 1
                if member not in decls.structs[stype]:
                    problem(off + m.start(), f"{var}.{m.group(1)}: {stype} has no member '{member}'")
                    continue
                # one level deeper: x.a.b where a is itself a struct typedef
                mtype = decls.structs[stype].get(member)
                rest = re.match(r"(?:\[[^\]]*\])?\.([A-Za-z_]\w*)", body[m.end():])
                if rest and mtype in decls.structs and _member_name(rest.group(1)) not in decls.structs[mtype]:
                    problem(off + m.start(), f"{var}.{m.group(1)}.{rest.group(1)}: {mtype} has no member '{_member_name(rest.group(1))}'")


def package_decls(pkg_dir, include_dirs):
    """C declarations visible to the cgo files of one package directory."""
    d = CDecls()
    seen = set()

    def pull(header):
        for base in [pkg_dir] + list(include_dirs):
            p = os.path.join(base, header)
            if os.path.exists(p) and p not in seen:
                seen.add(p)
                sub = parse_c(open(p).read())
                d.merge(sub)
                for inc in sub.includes:
                    pull(inc)
                return

    for f in sorted(os.listdir(pkg_dir)):
        if f.endswith(".go"):
            pre = cgo_preamble(open(os.path.join(pkg_dir, f)).read())
            if pre:
                sub = parse_c(pre)
                d.merge(sub)
                for inc in sub.includes:
                    pull(inc)
    return d


def exported_names(pkg_dir):
    names = set()
    for f in sorted(os.listdir(pkg_dir)):
        if not f.endswith(".go") or f.endswith("_test.go"):
            continue
        code = strip_go(open(os.path.join(pkg_dir, f)).read())
        names |= set(re.findall(r"^func\s+([A-Z]\w*)\s*\(", code, flags=re.M))
        names |= set(re.findall(r"^type\s+([A-Z]\w*)\b", code, flags=re.M))
        names |= set(re.findall(r"^(?:var|const)\s+([A-Z]\w*)\b", code, flags=re.M))
        for m in re.finditer(r"^(?:var|const|type)\s*\(", code, flags=re.M):
            end = _matching(code, m.end() - 1)
            block, depth = code[m.end():end], 0
            for line in block.split("\n"):
                if depth == 0:
                    for name in re.findall(r"^\s*([A-Z]\w*)(?:\s*,\s*([A-Z]\w*))*\b", line):
                        names |= {n for n in name if n}
                depth += line.count("{") + line.count("(") - line.count("}") - line.count(")")
    return names


def check_tree(go_root, include_dirs, stats=None):
    """Every problem found under go_root (the directory holding go.mod), as 'file:line: message' strings.
    stats (a dict) receives how much was looked at, so that a caller can tell a clean tree from a blind check."""
    problems = []
    stats = stats if stats is not None else defaultdict(int)
    module = None
    gomod = os.path.join(go_root, "go.mod")
    if os.path.exists(gomod):
        m = re.search(r"^module\s+(\S+)", open(gomod).read(), flags=re.M)
        module = m.group(1) if m else None
    else:
        problems.append(f"{gomod}: missing")
    exports = {}
    for dirpath, _dirs, files in sorted(os.walk(go_root)):
        gofiles = sorted(f for f in files if f.endswith(".go"))
        if not gofiles:
            continue
        decls = package_decls(dirpath, include_dirs)
        pkgs = set()
        defined = defaultdict(list)
        for f in gofiles:
            path = os.path.join(dirpath, f)
            code = strip_go(open(path).read())
            m = re.search(r"^package\s+(\w+)", code, flags=re.M)
            if m:
                pkgs.add(m.group(1)[:-5] if m.group(1).endswith("_test") else m.group(1))
            for name in re.findall(r"^func\s+([A-Za-z_]\w*)\s*\(", code, flags=re.M):
                if name not in ("init", "main") or name == "main":
                    defined[name].append(f)
            for name in re.findall(r"^type\s+([A-Za-z_]\w*)\b", code, flags=re.M):
                defined[name].append(f)
            check_go_file(path, decls, problems, stats)
            stats["go_files"] += 1
            for name in re.findall(r"^//export\s+(\w+)", open(path).read(), flags=re.M):
                if not re.search(r"^func\s+%s\s*\(" % name, code, flags=re.M):
                    problems.append(f"{path}: //export {name} without a Go function of that name")
                decls.functions.setdefault(name, None)
        if len(pkgs) > 1:
            problems.append(f"{dirpath}: files declare different packages: {sorted(pkgs)}")
        for name, where in defined.items():
            if len(where) > 1 and name != "init":
                problems.append(f"{dirpath}: {name} is defined in {' and '.join(where)}")
        for f in sorted(x for x in files if x.endswith(".c")):
            ctext = strip_c_comments(open(os.path.join(dirpath, f)).read())
            cdefs = parse_c(ctext)
            for m in re.finditer(r"\b([A-Za-z_]\w*)\s*\(", ctext):
                name = m.group(1)
                if name in ("if", "for", "while", "switch", "return", "sizeof") or name in cdefs.functions:
                    continue
                if re.match(r"(gm_|he_|go[A-Z])", name) and name not in decls.functions:
                    problems.append(f"{os.path.join(dirpath, f)}: {name} is not declared")
        if module:
            exports[module + "/" + os.path.relpath(dirpath, go_root).replace(os.sep, "/")] = exported_names(dirpath)
    # samples and tests against the exported surface of the binding packages
    for dirpath, _dirs, files in sorted(os.walk(go_root)):
        for f in sorted(x for x in files if x.endswith(".go")):
            path = os.path.join(dirpath, f)
            raw = open(path).read()
            code = strip_go(raw)
            for name, ipath in _imports(code, raw):
                if ipath not in exports:
                    if module and ipath.startswith(module):
                        problems.append(f"{path}: imports {ipath}, which is not a package of this module")
                    continue
                for m in re.finditer(r"(?<![\w\.])%s\.([A-Za-z_]\w*)" % re.escape(name), code):
                    stats["package_references_checked"] += 1
                    if m.group(1) not in exports[ipath]:
                        problems.append(f"{path}:{code.count(chr(10), 0, m.start()) + 1}: {name}.{m.group(1)} is not exported by {ipath}")
    return problems
