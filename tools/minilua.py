#!/usr/bin/env python3
"""A small interpreter for the subset of Lua 5.1 that game scripts like the reference's game.lua
use.  TEST INFRASTRUCTURE: it exists so that the script's own text can be executed in a
container without Lua, to produce known answers for the restatements of its logic
(pwnfps_amd/script.py, host/game_script.c) -- tools/gen_script_golden.py.  It knows nothing
about the game: it parses and runs Lua.

Lua numbers are C doubles and math.sin / cos / fmod / floor ... are the C library's; Python floats
are the same doubles and Python's math module calls the same libm, so arithmetic is the VM's.

Supported: nil / booleans / numbers / strings, tables (constructors with positional, [k] = v and
name = v fields; # on sequences), locals and globals, multiple assignment, arithmetic (+ - * / % ^,
unary -), comparison, and / or / not, .. , function definitions (global, local, anonymous, with
closures), calls, return (multiple values), if / elseif / else, while, repeat-until, numeric and
generic for (pairs, ipairs), break, -- and --[[ ]] comments; math.*, string.format / sub / len /
rep, tostring, tonumber, print, type, select('#'), unpack, assert, error.  Not supported:
metatables, coroutines, goto, varargs beyond select / ..., integer division, the rest of the
library.
"""
import math
import re

TOKEN = re.compile(r"""
    (?P<ws>\s+) | (?P<longcomment>--\[\[.*?\]\]) | (?P<comment>--[^\n]*) |
    (?P<number>0[xX][0-9a-fA-F]+ | \d+\.?\d*(?:[eE][+-]?\d+)? | \.\d+(?:[eE][+-]?\d+)?) |
    (?P<name>[A-Za-z_][A-Za-z_0-9]*) |
    (?P<string>"(?:\\.|[^"\\])*" | '(?:\\.|[^'\\])*') |
    (?P<longstring>\[\[.*?\]\]) |
    (?P<op>\.\.\.|\.\.|==|~=|<=|>=|[-+*/%^#<>=(){}\[\];:,.])
""", re.X | re.S)
KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "if", "in", "local", "nil",
            "not", "or", "repeat", "return", "then", "true", "until", "while"}


class LuaError(Exception):
    pass


def lex(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if m is None:
            raise LuaError("unexpected character %r at offset %d" % (src[pos], pos))
        pos = m.end()
        kind = m.lastgroup
        if kind in ("ws", "comment", "longcomment"):
            continue
        text = m.group()
        if kind == "number":
            out.append(("number", float(int(text, 16)) if text[:2].lower() == "0x" else float(text)))
        elif kind == "name":
            out.append(("kw", text) if text in KEYWORDS else ("name", text))
        elif kind == "string":
            body = text[1:-1]
            body = re.sub(r"\\(\d{1,3}|.)", lambda q: {"n": "\n", "t": "\t", "\\": "\\", '"': '"', "'": "'", "r": "\r", "0": "\0"}.get(
                q.group(1), chr(int(q.group(1))) if q.group(1).isdigit() else q.group(1)), body)
            out.append(("string", body))
        elif kind == "longstring":
            out.append(("string", text[2:-2]))
        else:
            out.append(("op", text))
    out.append(("eof", None))
    return out


# ---------------------------------------------------------------- parser ----
BINPRI = {"or": (1, 1), "and": (2, 2), "<": (3, 3), ">": (3, 3), "<=": (3, 3), ">=": (3, 3), "~=": (3, 3), "==": (3, 3),
          "..": (5, 4), "+": (6, 6), "-": (6, 6), "*": (7, 7), "/": (7, 7), "%": (7, 7), "^": (10, 9)}
UNARY_PRI = 8


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def check(self, kind, val=None):
        k, v = self.t[self.i]
        return k == kind and (val is None or v == val)

    def accept(self, kind, val=None):
        if self.check(kind, val):
            return self.next()
        return None

    def expect(self, kind, val=None):
        tok = self.accept(kind, val)
        if tok is None:
            raise LuaError("expected %s %r, got %r" % (kind, val, self.peek()))
        return tok

    def block(self):
        stmts = []
        while not (self.check("eof") or (self.peek()[0] == "kw" and self.peek()[1] in ("end", "else", "elseif", "until"))):
            if self.accept("op", ";"):
                continue
            if self.check("kw", "return"):
                self.next()
                vals = [] if (self.check("eof") or self.check("op", ";") or (self.peek()[0] == "kw" and self.peek()[1] in ("end", "else", "elseif", "until"))) else self.exprlist()
                self.accept("op", ";")
                stmts.append(("return", vals))
                break
            stmts.append(self.statement())
        return stmts

    def statement(self):
        k, v = self.peek()
        if k == "kw":
            if v == "if":
                self.next()
                arms = []
                cond = self.expr()
                self.expect("kw", "then")
                arms.append((cond, self.block()))
                other = None
                while True:
                    if self.accept("kw", "elseif"):
                        cond = self.expr()
                        self.expect("kw", "then")
                        arms.append((cond, self.block()))
                    elif self.accept("kw", "else"):
                        other = self.block()
                        self.expect("kw", "end")
                        break
                    else:
                        self.expect("kw", "end")
                        break
                return ("if", arms, other)
            if v == "while":
                self.next()
                cond = self.expr()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("while", cond, body)
            if v == "repeat":
                self.next()
                body = self.block()
                self.expect("kw", "until")
                return ("repeat", body, self.expr())
            if v == "do":
                self.next()
                body = self.block()
                self.expect("kw", "end")
                return ("do", body)
            if v == "for":
                self.next()
                n1 = self.expect("name")[1]
                if self.accept("op", "="):
                    a = self.expr()
                    self.expect("op", ",")
                    b = self.expr()
                    c = self.expr() if self.accept("op", ",") else None
                    self.expect("kw", "do")
                    body = self.block()
                    self.expect("kw", "end")
                    return ("fornum", n1, a, b, c, body)
                names = [n1]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                self.expect("kw", "in")
                exprs = self.exprlist()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("forin", names, exprs, body)
            if v == "function":
                self.next()
                target = ("name", self.expect("name")[1])
                while self.accept("op", "."):
                    target = ("index", target, ("const", self.expect("name")[1]))
                return ("assign", [target], [self.funcbody()])
            if v == "local":
                self.next()
                if self.accept("kw", "function"):
                    name = self.expect("name")[1]
                    return ("localfunc", name, self.funcbody())
                names = [self.expect("name")[1]]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                vals = self.exprlist() if self.accept("op", "=") else []
                return ("local", names, vals)
            if v == "break":
                self.next()
                return ("break",)
        e = self.suffixed()
        if self.check("op", "=") or self.check("op", ","):
            targets = [e]
            while self.accept("op", ","):
                targets.append(self.suffixed())
            self.expect("op", "=")
            for tg in targets:
                if tg[0] not in ("name", "index"):
                    raise LuaError("cannot assign to %r" % (tg,))
            return ("assign", targets, self.exprlist())
        if e[0] != "call":
            raise LuaError("syntax error near %r" % (self.peek(),))
        return ("callstat", e)

    def funcbody(self):
        self.expect("op", "(")
        params, vararg = [], False
        if not self.check("op", ")"):
            while True:
                if self.accept("op", "..."):
                    vararg = True
                    break
                params.append(self.expect("name")[1])
                if not self.accept("op", ","):
                    break
        self.expect("op", ")")
        body = self.block()
        self.expect("kw", "end")
        return ("function", params, vararg, body)

    def exprlist(self):
        out = [self.expr()]
        while self.accept("op", ","):
            out.append(self.expr())
        return out

    def primary(self):
        k, v = self.next()
        if k == "name":
            return ("name", v)
        if k == "op" and v == "(":
            e = self.expr()
            self.expect("op", ")")
            return ("paren", e)
        raise LuaError("unexpected %r" % ((k, v),))

    def suffixed(self):
        e = self.primary()
        while True:
            if self.accept("op", "."):
                e = ("index", e, ("const", self.expect("name")[1]))
            elif self.accept("op", "["):
                k = self.expr()
                self.expect("op", "]")
                e = ("index", e, k)
            elif self.check("op", "("):
                self.next()
                args = [] if self.check("op", ")") else self.exprlist()
                self.expect("op", ")")
                e = ("call", e, args)
            elif self.check("string"):
                e = ("call", e, [("const", self.next()[1])])
            elif self.check("op", "{"):
                e = ("call", e, [self.table()])
            else:
                return e

    def table(self):
        self.expect("op", "{")
        arr, rec = [], []
        while not self.check("op", "}"):
            if self.check("op", "["):
                self.next()
                k = self.expr()
                self.expect("op", "]")
                self.expect("op", "=")
                rec.append((k, self.expr()))
            elif self.check("name") and self.t[self.i + 1] == ("op", "="):
                k = ("const", self.next()[1])
                self.next()
                rec.append((k, self.expr()))
            else:
                arr.append(self.expr())
            if not (self.accept("op", ",") or self.accept("op", ";")):
                break
        self.expect("op", "}")
        return ("table", arr, rec)

    def simple(self):
        k, v = self.peek()
        if k == "number" or k == "string":
            self.next()
            return ("const", v)
        if k == "kw" and v in ("nil", "true", "false"):
            self.next()
            return ("const", {"nil": None, "true": True, "false": False}[v])
        if k == "op" and v == "...":
            self.next()
            return ("vararg",)
        if k == "op" and v == "{":
            return self.table()
        if k == "kw" and v == "function":
            self.next()
            return self.funcbody()
        return self.suffixed()

    def expr(self, limit=0):
        k, v = self.peek()
        if (k == "kw" and v == "not") or (k == "op" and v in ("-", "#")):
            self.next()
            left = ("unop", v, self.expr(UNARY_PRI))
        else:
            left = self.simple()
        while True:
            k, v = self.peek()
            if not ((k == "op" or k == "kw") and v in BINPRI):
                return left
            lp, rp = BINPRI[v]
            if lp <= limit:
                return left
            self.next()
            left = ("binop", v, left, self.expr(rp))


# ------------------------------------------------------------- interpreter ----
class LuaTable:
    def __init__(self):
        self.h = {}

    def get(self, k):
        if isinstance(k, float) and k.is_integer():
            k = int(k)
        return self.h.get(k)

    def set(self, k, v):
        if k is None:
            raise LuaError("table index is nil")
        if isinstance(k, float) and k.is_integer():
            k = int(k)
        if v is None:
            self.h.pop(k, None)
        else:
            self.h[k] = v

    def length(self):
        n = 0
        while (n + 1) in self.h:
            n += 1
        return n


class Break(Exception):
    pass


class Return(Exception):
    def __init__(self, vals):
        self.vals = vals


class Scope:
    __slots__ = ("vars", "parent")

    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def find(self, name):
        s = self
        while s is not None:
            if name in s.vars:
                return s
            s = s.parent
        return None


class LuaFunction:
    def __init__(self, interp, params, vararg, body, scope):
        self.interp, self.params, self.vararg, self.body, self.scope = interp, params, vararg, body, scope

    def __call__(self, *args):
        sc = Scope(self.scope)
        for i, p in enumerate(self.params):
            sc.vars[p] = args[i] if i < len(args) else None
        if self.vararg:
            sc.vars["..."] = list(args[len(self.params):])
        try:
            self.interp.exec_block(self.body, sc)
        except Return as r:
            return r.vals
        return []


def truthy(v):
    return v is not None and v is not False


def tostr(v):
    if v is None:
        return "nil"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, float):
        return "%.14g" % v
    return str(v)


def tonum(v):
    if isinstance(v, float):
        return v
    if isinstance(v, str):
        try:
            return float(int(v, 16)) if v.strip().lower().startswith("0x") else float(v)
        except ValueError:
            return None
    return None


class Interp:
    def __init__(self):
        self.globals = LuaTable()
        m = LuaTable()
        for name in ("sin", "cos", "tan", "asin", "acos", "atan", "sqrt", "exp", "fabs", "ceil"):
            if hasattr(math, name):
                m.set(name, (lambda f: lambda x, *r: [float(f(x))])(getattr(math, name)))
        m.set("abs", lambda x, *r: [abs(x)])
        m.set("floor", lambda x, *r: [float(math.floor(x))])
        m.set("fmod", lambda a, b, *r: [math.fmod(a, b)])
        m.set("atan2", lambda a, b, *r: [math.atan2(a, b)])
        m.set("pow", lambda a, b, *r: [math.pow(a, b)])
        m.set("log", lambda a, *r: [math.log(a)])
        m.set("max", lambda *a: [max(a)])
        m.set("min", lambda *a: [min(a)])
        m.set("pi", math.pi)
        m.set("huge", math.inf)
        self.globals.set("math", m)
        s = LuaTable()
        s.set("format", lambda fmt, *a: [self._format(fmt, a)])
        s.set("sub", lambda st, i, j=-1.0, *r: [self._sub(st, int(i), int(j))])
        s.set("len", lambda st, *r: [float(len(st))])
        s.set("rep", lambda st, n, *r: [st * int(n)])
        self.globals.set("string", s)
        self.out = []
        self.globals.set("print", lambda *a: self.out.append("\t".join(tostr(x) for x in a)) or [])
        self.globals.set("tostring", lambda v=None, *r: [tostr(v)])
        self.globals.set("tonumber", lambda v=None, *r: [tonum(v)])
        self.globals.set("type", lambda v=None, *r: ["nil" if v is None else "boolean" if isinstance(v, bool) else "number" if isinstance(v, float)
                                                     else "string" if isinstance(v, str) else "table" if isinstance(v, LuaTable) else "function"])
        self.globals.set("ipairs", lambda t, *r: [self._ipairs_next, t, 0.0])
        self.globals.set("pairs", lambda t, *r: [self._pairs_iter(t), t, None])
        self.globals.set("unpack", lambda t, *r: [t.get(i) for i in range(1, t.length() + 1)])
        self.globals.set("select", lambda n, *a: [float(len(a))] if n == "#" else list(a[int(n) - 1:]))
        self.globals.set("assert", self._assert)
        self.globals.set("error", self._error)

    @staticmethod
    def _format(fmt, args):
        args = list(args)

        def one(m):
            if m.group(0) == "%%":
                return "%"
            v = args.pop(0) if args else None
            conv = m.group(0)[-1]
            if conv in "dixXco":
                v = int(tonum(v))
            elif conv in "eEfgG":
                v = tonum(v)
            elif conv == "s":
                v = tostr(v)
            return m.group(0) % v
        return re.sub(r"%[-+ 0#]*\d*(?:\.\d+)?[dixXcoeEfgGs%]", one, fmt)

    @staticmethod
    def _sub(st, i, j):
        n = len(st)
        if i < 0:
            i = max(n + i + 1, 1)
        if j < 0:
            j = n + j + 1
        return st[max(i, 1) - 1:min(j, n)]

    @staticmethod
    def _ipairs_next(t, i, *r):
        v = t.get(i + 1)
        return [None] if v is None else [i + 1, v]

    @staticmethod
    def _pairs_iter(t):
        keys = list(t.h.keys())
        pos = {"i": 0}

        def nxt(tt, k=None, *r):
            while pos["i"] < len(keys):
                kk = keys[pos["i"]]
                pos["i"] += 1
                if kk in tt.h:
                    return [float(kk) if isinstance(kk, int) else kk, tt.h[kk]]
            return [None]
        return nxt

    @staticmethod
    def _assert(v=None, msg="assertion failed!", *r):
        if not truthy(v):
            raise LuaError(tostr(msg))
        return [v]

    @staticmethod
    def _error(msg=None, *r):
        raise LuaError(tostr(msg))

    # -- registration of host functions: f(*args) -> value, tuple/list of values or None
    def register(self, name, fn):
        def wrap(*a):
            r = fn(*a)
            if r is None:
                return []
            return list(r) if isinstance(r, (tuple, list)) else [r]
        self.globals.set(name, wrap)

    def run(self, src):
        self.exec_block(Parser(lex(src)).block(), None_scope(self))

    def call(self, name, *args):
        f = self.globals.get(name)
        if f is None:
            raise LuaError("attempt to call global '%s' (a nil value)" % name)
        return f(*[float(a) if isinstance(a, (int, float)) and not isinstance(a, bool) else a for a in args])

    # -- statements
    def exec_block(self, stmts, scope):
        sc = Scope(scope)
        for st in stmts:
            self.exec(st, sc)

    def exec(self, st, sc):
        op = st[0]
        if op == "local":
            vals = self.evallist(st[2], sc)
            for i, n in enumerate(st[1]):
                sc.vars[n] = vals[i] if i < len(vals) else None
        elif op == "assign":
            vals = self.evallist(st[2], sc)
            # (targets' table and key expressions are evaluated before any store, like the VM does)
            refs = []
            for tg in st[1]:
                refs.append((self.eval(tg[1], sc), self.eval(tg[2], sc)) if tg[0] == "index" else None)
            for i, tg in enumerate(st[1]):
                v = vals[i] if i < len(vals) else None
                if tg[0] == "name":
                    s = sc.find(tg[1]) if sc is not None else None
                    if s is not None:
                        s.vars[tg[1]] = v
                    else:
                        self.globals.set(tg[1], v)
                else:
                    t, k = refs[i]
                    if not isinstance(t, LuaTable):
                        raise LuaError("attempt to index a %s value" % tostr(t))
                    t.set(k, v)
        elif op == "callstat":
            self.eval_multi(st[1], sc)
        elif op == "if":
            for cond, body in st[1]:
                if truthy(self.eval(cond, sc)):
                    self.exec_block(body, sc)
                    return
            if st[2] is not None:
                self.exec_block(st[2], sc)
        elif op == "fornum":
            a, b = self.eval(st[2], sc), self.eval(st[3], sc)
            c = self.eval(st[4], sc) if st[4] is not None else 1.0
            i = a
            try:
                while (c > 0 and i <= b) or (c <= 0 and i >= b):
                    inner = Scope(sc)
                    inner.vars[st[1]] = i
                    self.exec_block(st[5], inner)
                    i = i + c
            except Break:
                pass
        elif op == "forin":
            vals = self.evallist(st[2], sc)
            f, s, ctl = (vals + [None, None, None])[:3]
            try:
                while True:
                    r = f(s, ctl)
                    if not r or r[0] is None:
                        break
                    ctl = r[0]
                    inner = Scope(sc)
                    for i, n in enumerate(st[1]):
                        inner.vars[n] = r[i] if i < len(r) else None
                    self.exec_block(st[3], inner)
            except Break:
                pass
        elif op == "while":
            try:
                while truthy(self.eval(st[1], sc)):
                    self.exec_block(st[2], sc)
            except Break:
                pass
        elif op == "repeat":
            try:
                while True:
                    inner = Scope(sc)
                    for s2 in st[1]:
                        self.exec(s2, inner)
                    if truthy(self.eval(st[2], inner)):
                        break
            except Break:
                pass
        elif op == "do":
            self.exec_block(st[1], sc)
        elif op == "localfunc":
            sc.vars[st[1]] = None
            sc.vars[st[1]] = self.eval(st[2], sc)
        elif op == "return":
            raise Return(self.evallist(st[1], sc))
        elif op == "break":
            raise Break()
        else:
            raise LuaError("unknown statement %r" % (op,))

    # -- expressions
    def evallist(self, exprs, sc):
        out = []
        for i, e in enumerate(exprs):
            if i == len(exprs) - 1 and e[0] in ("call", "vararg"):
                out.extend(self.eval_multi(e, sc))
            else:
                out.append(self.eval(e, sc))
        return out

    def eval_multi(self, e, sc):
        if e[0] == "call":
            f = self.eval(e[1], sc)
            args = self.evallist(e[2], sc)
            if f is None or isinstance(f, (float, str, bool, LuaTable)):
                raise LuaError("attempt to call a %s value" % tostr(f))
            r = f(*args)
            return list(r) if r is not None else []
        if e[0] == "vararg":
            s = sc.find("...")
            return list(s.vars["..."]) if s is not None else []
        return [self.eval(e, sc)]

    def eval(self, e, sc):
        op = e[0]
        if op == "const":
            return e[1]
        if op == "name":
            s = sc.find(e[1]) if sc is not None else None
            return s.vars[e[1]] if s is not None else self.globals.get(e[1])
        if op == "index":
            t = self.eval(e[1], sc)
            k = self.eval(e[2], sc)
            if isinstance(t, LuaTable):
                return t.get(k)
            if isinstance(t, str):
                return self.globals.get("string").get(k)
            raise LuaError("attempt to index a %s value" % ("nil" if t is None else type(t).__name__))
        if op in ("call", "vararg"):
            r = self.eval_multi(e, sc)
            return r[0] if r else None
        if op == "paren":
            return self.eval(e[1], sc)
        if op == "function":
            return LuaFunction(self, e[1], e[2], e[3], sc)
        if op == "table":
            t = LuaTable()
            n = 0
            for i, a in enumerate(e[1]):
                if i == len(e[1]) - 1 and a[0] in ("call", "vararg"):
                    for v in self.eval_multi(a, sc):
                        n += 1
                        t.set(float(n), v)
                else:
                    n += 1
                    t.set(float(n), self.eval(a, sc))
            for k, v in e[2]:
                t.set(self.eval(k, sc), self.eval(v, sc))
            return t
        if op == "unop":
            v = self.eval(e[2], sc)
            if e[1] == "not":
                return not truthy(v)
            if e[1] == "-":
                n = tonum(v)
                if n is None:
                    raise LuaError("attempt to perform arithmetic on a %s value" % tostr(v))
                return -n
            if isinstance(v, str):
                return float(len(v))
            if isinstance(v, LuaTable):
                return float(v.length())
            raise LuaError("attempt to get length of a %s value" % tostr(v))
        if op == "binop":
            o = e[1]
            if o == "and":
                a = self.eval(e[2], sc)
                return self.eval(e[3], sc) if truthy(a) else a
            if o == "or":
                a = self.eval(e[2], sc)
                return a if truthy(a) else self.eval(e[3], sc)
            a, b = self.eval(e[2], sc), self.eval(e[3], sc)
            if o == "==":
                return a is b if isinstance(a, (LuaTable, LuaFunction)) else (type(a) is type(b) or (isinstance(a, float) and isinstance(b, float))) and a == b
            if o == "~=":
                return not (a is b if isinstance(a, (LuaTable, LuaFunction)) else (type(a) is type(b) or (isinstance(a, float) and isinstance(b, float))) and a == b)
            if o == "..":
                if not isinstance(a, (str, float)) or not isinstance(b, (str, float)):
                    raise LuaError("attempt to concatenate a %s value" % tostr(a if not isinstance(a, (str, float)) else b))
                return tostr(a) + tostr(b)
            if o in ("<", "<=", ">", ">="):
                if not ((isinstance(a, float) and isinstance(b, float)) or (isinstance(a, str) and isinstance(b, str))):
                    raise LuaError("attempt to compare %s with %s" % (tostr(a), tostr(b)))
                return {"<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[o]
            x, y = tonum(a), tonum(b)
            if x is None or y is None:
                raise LuaError("attempt to perform arithmetic on a %s value" % tostr(a if x is None else b))
            if o == "+":
                return x + y
            if o == "-":
                return x - y
            if o == "*":
                return x * y
            if o == "/":
                if y == 0.0:
                    return math.nan if x == 0.0 or x != x else math.copysign(math.inf, x) * math.copysign(1.0, y)
                return x / y
            if o == "%":
                # luai_nummod: a - floor(a/b)*b
                return x - math.floor(x / y) * y if y != 0.0 else math.nan
            if o == "^":
                return math.pow(x, y)
        raise LuaError("unknown expression %r" % (op,))


def None_scope(interp):
    return None
