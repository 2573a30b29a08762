#!/usr/bin/env python3
"""Writes tests/golden/normalize_vectors.json: what Blurrily::Map#normalize_string (lib/blurrily/map.rb:40-47)
makes of non-ASCII needles, FROZEN.

The reference decomposes with ActiveSupport 4.2's NFKD tables (Gemfile.lock:11), which are not under
/root/reference; the product uses Python's unicodedata.  Neither is consulted here: the expected strings are
worked out from the literal table below -- the Unicode Character Database's decomposition mappings of the Latin-1
Supplement, Latin Extended-A, the Latin ligatures and a few compatibility forms, which Unicode's stability
policy forbids to change -- following the Ruby text step by step:

    downcase (ASCII only on the Rubies the reference supports) -> NFKD -> delete non-ASCII ->
    every character outside [a-z] becomes a space -> squeeze whitespace -> strip

so that a Python or Unicode upgrade that changed the product's behaviour on these inputs fails
tests/test_normalize.py.  (What ActiveSupport's own tables say stays unpinned: DESIGN.md section 6.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# character -> the ASCII characters its NFKD decomposition contains (combining marks and other non-ASCII
# characters of the decomposition are deleted by the next step of map.rb:44 anyway); "" = nothing ASCII left
ASCII_OF = {}


def put(chars, ascii_part):
    for c in chars:
        ASCII_OF[c] = ascii_part


# Latin-1 Supplement (U+00A0..U+00FF)
put(" ", " ")                      # NO-BREAK SPACE <noBreak> 0020
put("¨¯´¸", " ")    # spacing diaeresis / macron / acute / cedilla: <compat> 0020 + a mark
put("ª", "a"); put("º", "o")  # feminine / masculine ordinal
put("²", "2"); put("³", "3"); put("¹", "1")
put("¼", "14"); put("½", "12"); put("¾", "34")     # vulgar fractions: digit, U+2044, digit
put("µ", "")                       # MICRO SIGN -> GREEK SMALL LETTER MU
put("ÀÁÂÃÄÅ", "A"); put("Ç", "C"); put("ÈÉÊË", "E")
put("ÌÍÎÏ", "I"); put("Ñ", "N"); put("ÒÓÔÕÖ", "O")
put("ÙÚÛÜ", "U"); put("Ý", "Y")
put("àáâãäå", "a"); put("ç", "c"); put("èéêë", "e")
put("ìíîï", "i"); put("ñ", "n"); put("òóôõö", "o")
put("ùúûü", "u"); put("ýÿ", "y")
# no decomposition at all: deleted as non-ASCII
put("¡¢£¤¥¦§©«¬­®°±¶·»¿", "")
put("ÆÐ×ØÞßæð÷øþ", "")   # AE ETH x O-stroke THORN sharp-s ...
# Latin Extended-A (small letters; the capitals decompose to ASCII capitals, which then become spaces)
put("āăą", "a"); put("ćĉċč", "c"); put("ď", "d"); put("đ", "")
put("ēĕėęě", "e"); put("ĝğġģ", "g"); put("ĥ", "h"); put("ħ", "")
put("ĩīĭį", "i"); put("ı", ""); put("ĳ", "ij"); put("ĵ", "j"); put("ķ", "k")
put("ĺļľŀ", "l"); put("ł", ""); put("ńņňŉ", "n")
put("ōŏő", "o"); put("œ", ""); put("ŕŗř", "r"); put("śŝşš", "s")
put("ţť", "t"); put("ŧ", ""); put("ũūŭůűų", "u"); put("ŵ", "w")
put("ŷ", "y"); put("źżž", "z"); put("ſ", "s")
put("ĀĂĄ", "A"); put("Ł", ""); put("İ", "I"); put("Š", "S"); put("Ž", "Z"); put("Ĳ", "IJ")
# Latin ligatures and fullwidth forms (compatibility decompositions)
put("ﬀ", "ff"); put("ﬁ", "fi"); put("ﬂ", "fl"); put("ﬃ", "ffi"); put("ﬄ", "ffl")
put("ﬅﬆ", "st"); put("ａ", "a"); put("ｚ", "z"); put("Ａ", "A")
put("€", ""); put("’", ""); put("–", "")          # euro sign, right quote, en dash: no ASCII


def expected(raw):
    s = "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in raw)          # ASCII downcase (map.rb:41)
    lines = s.split("\n")
    if any(ln and all(("a" <= c <= "z") or c == " " for c in ln) for ln in lines):   # /^([a-z ])+$/ (map.rb:42)
        pass
    else:
        s = "".join(ASCII_OF[c] if ord(c) > 127 else c for c in s)               # NFKD + delete non-ASCII (:43-44)
        s = "".join(c if "a" <= c <= "z" else " " for c in s)                    # [^a-z] -> ' ' (:45)
    out = " ".join(s.replace("\t", " ").replace("\n", " ").replace("\r", " ").replace("\f", " ").replace("\v", " ").split(" "))
    while "  " in out:
        out = out.replace("  ", " ")
    return out.strip(" ")


WORDS = ["zürich", "Zürich", "são paulo", "SÃO PAULO", "łódź", "kraków", "málaga",
         "ærø", "straße", "münchen", "İstanbul", "ıstanbul", "besançon", "côte d’ivoire",
         "saint-étienne", "Saint-Étienne", "tromsø", "göteborg", "ålesund", "reykjavík", "plaža",
         "české budějovice", "ștefan", "naïve café", "ﬁsh ﬂower oﬃce", "ａｚ Ａ",
         "50µm", "1½ cups", "nº 5", "a b", "coöperate–now", "@€%é", "Ĳsselmeer ĳsselmeer",
         "lŀl ŉ ſt", "déjà vu\n%%", "plain line\néè"]
WORDS.remove("ștefan")             # (U+0219 is Latin Extended-B: outside the table)


def main():
    vectors = [{"raw": w, "normalized": expected(w)} for w in WORDS]
    # every character of the table on its own, between two letters
    for c in sorted(ASCII_OF):
        raw = "x" + c + "y"
        vectors.append({"raw": raw, "normalized": expected(raw)})
    out = {"what": " ".join(__doc__.split("\n\n")[0].split()), "source": "tools/make_normalize_vectors.py (literal UCD table, not unicodedata)",
           "vectors": vectors}
    path = os.path.join(ROOT, "tests", "golden", "normalize_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=True)
    print(f"wrote {path}: {len(vectors)} vectors")


if __name__ == "__main__":
    main()
