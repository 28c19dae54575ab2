"""tools/minilua.py, the interpreter that runs the reference's game.lua for the script fixture,
is generic Lua: known answers of the language itself (Lua 5.1 reference manual), nothing of the game."""
import math
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import minilua  # noqa: E402


def run(src, **host):
    L = minilua.Interp()
    for k, v in host.items():
        L.register(k, v)
    L.run(src)
    return L


def g(L, name):
    return L.globals.get(name)


def test_arithmetic_precedence_and_number_semantics():
    L = run("""
        a = 2 + 3 * 4 ^ 2 / 8          -- ^ binds tighter than unary minus and * /
        b = -2 ^ 2
        c = 2 ^ 3 ^ 2                  -- right associative
        d = 7 % 3; e = -7 % 3; f = 7 % -3; m = math.fmod(-7, 3)     -- % is floored, fmod truncates
        h = 1 / 0; i = -1 / 0
        j = 10 / 4
        k = "10" + 5                   -- string coerced to number
        l = 1 .. 2                     -- numbers concatenate as strings
    """)
    assert g(L, "a") == 8.0 and g(L, "b") == -4.0 and g(L, "c") == 512.0
    assert (g(L, "d"), g(L, "e"), g(L, "f"), g(L, "m")) == (1.0, 2.0, -2.0, -1.0)
    assert g(L, "h") == math.inf and g(L, "i") == -math.inf and g(L, "j") == 2.5
    assert g(L, "k") == 15.0 and g(L, "l") == "12"


def test_logic_comparison_and_truthiness():
    L = run("""
        a = nil or 5; b = false and 1; c = 0 and "zero is true"; d = not nil; e = not 0
        f = 1 < 2 and "yes" or "no"
        g1 = "a" < "b"; h = 1 == 1.0; i = "1" == 1; j = 2 ~= 3
        k = nil == false
    """)
    assert g(L, "a") == 5.0 and g(L, "b") is False and g(L, "c") == "zero is true"
    assert g(L, "d") is True and g(L, "e") is False and g(L, "f") == "yes"
    assert g(L, "g1") is True and g(L, "h") is True and g(L, "i") is False and g(L, "j") is True and g(L, "k") is False


def test_tables_length_and_multiple_assignment():
    L = run("""
        t = {10, 20, 30, x = 1, [5] = 50}
        n = #t
        t[#t + 1] = 40
        n2 = #t
        a, b, c = 1, 2                 -- missing values are nil
        x, y = 1, 2
        x, y = y, x                    -- right-hand sides are evaluated before any store
        p = {{1, 2}, {3, 4}}
        p[1][2], p[2][1] = p[2][1], p[1][2]
        q = p[1][2] * 10 + p[2][1]
        i = 1
        u = {}
        i, u[i] = i + 1, 20            -- the index expression is evaluated before the assignment
    """)
    assert g(L, "n") == 3.0 and g(L, "n2") == 5.0          # t[5] was there: 5 is the only border now
    assert g(L, "a") == 1.0 and g(L, "b") == 2.0 and g(L, "c") is None
    assert g(L, "x") == 2.0 and g(L, "y") == 1.0 and g(L, "q") == 32.0
    assert g(L, "i") == 2.0 and g(L, "u").get(1.0) == 20.0 and g(L, "u").get(2.0) is None


def test_control_flow_functions_and_closures():
    L = run("""
        function fib(n) if n < 2 then return n else return fib(n - 1) + fib(n - 2) end end
        f10 = fib(10)
        s = 0
        for i = 1, 10, 3 do s = s + i end           -- 1 4 7 10
        for i = 3, 1, -1 do s = s * 10 + i end
        w = 0
        while true do w = w + 1; if w > 4 then break end end
        r = 0
        repeat local z = r + 1; r = z until z >= 3   -- the condition sees the body's locals
        local function counter()
            local c = 0
            return function() c = c + 1; return c end
        end
        c1, c2 = counter(), counter()
        c1(); c1()
        v1, v2 = c1(), c2()
        function mr() return 1, 2, 3 end
        t = {mr(), mr()}                             -- only the last call is expanded
        nt = #t
        a, b = (mr())                                -- parentheses truncate to one value
        keys = 0
        for k, v in pairs({a = 1, b = 2, 3}) do keys = keys + 1 end
        sum = 0
        for i, v in ipairs({5, 6, nil, 8}) do sum = sum + v end
        local shadow = 1
        do local shadow = 2 end
        sh = shadow
        if false then e = 1 elseif nil then e = 2 else e = 3 end
    """)
    assert g(L, "f10") == 55.0 and g(L, "s") == 22321.0 and g(L, "w") == 5.0 and g(L, "r") == 3.0
    assert g(L, "v1") == 3.0 and g(L, "v2") == 1.0 and g(L, "nt") == 4.0
    assert g(L, "a") == 1.0 and g(L, "b") is None and g(L, "keys") == 3.0 and g(L, "sum") == 11.0
    assert g(L, "sh") == 1.0 and g(L, "e") == 3.0


def test_library_and_host_functions():
    seen = []
    L = run("""
        a = math.floor(-0.5); b = math.floor(2.7); c = math.max(1, 9, 3); d = math.pi
        e = math.sin(1.25); f = math.cos(1.25); h = math.fmod(5.5, 0.5)
        s = string.format("%d-%s-%.2f", 3, "x", 1.5); n = #"hello"; u = string.sub("hello", 2, 4)
        r1, r2 = host(1, "two")
        print("a", 1, nil)
        -- a long comment --[[ inside ]] is skipped
        z = tostring(1e15) .. tostring(0.1)
    """, host=lambda x, y: (seen.append((x, y)) or (x + 1, y + "!")))
    assert g(L, "a") == -1.0 and g(L, "b") == 2.0 and g(L, "c") == 9.0 and g(L, "d") == math.pi
    assert g(L, "e") == math.sin(1.25) and g(L, "f") == math.cos(1.25) and g(L, "h") == math.fmod(5.5, 0.5)
    assert g(L, "s") == "3-x-1.50" and g(L, "n") == 5.0 and g(L, "u") == "ell"
    assert seen == [(1.0, "two")] and g(L, "r1") == 2.0 and g(L, "r2") == "two!"
    assert L.out == ["a\t1\tnil"] and g(L, "z") == "1e+150.1"
    assert L.call("host", 4, "x") == [5.0, "x!"]


def test_errors_are_lua_errors():
    for src in ("x = nil + 1", "t = nil; y = t.z", "f = 5; f()", "x = 1 < 'a'", "x = )"):
        with pytest.raises(minilua.LuaError):
            run(src)
