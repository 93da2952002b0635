"""Flag masks and feature-column indexes of the parse matrix.

Mirror of reference latok/core/offsets.py:3-49 (itself generated next to latok.h:3-49): the same 20 ``*_MASK`` names,
25 ``*_IDX`` names and ``FEATURE_COUNT``, with identical values, so callers that build combo matrices from
``oft.SPACE_IDX`` etc. keep working.  Values are produced from ordered name lists instead of being spelled out.
"""

_FLAG_NAMES = (
    "ALPHA", "DECIMAL", "DIGIT", "LOWER", "LINEBREAK", "SPACE", "TITLE", "UPPER", "XID_START", "XID_CONTINUE",
    "PRINTABLE", "NUMERIC", "CASE_IGNORABLE", "CASED", "EXTENDED_CASE", "SPECIALS", "CHAR_AT", "CHAR_COLON",
    "CHAR_SLASH", "CHAR_PERIOD",
)
_COLUMN_NAMES = (
    "ALPHA", "ALPHA_NUM", "NUM", "LOWER", "UPPER", "SPACE", "SYMBOL", "TWITTER", "CHAR_AT", "CHAR_COLON",
    "CHAR_SLASH", "CHAR_PERIOD", "PREV_ALPHA", "NEXT_ALPHA", "PREV_ALPHA_NUM", "NEXT_ALPHA_NUM", "PREV_LOWER",
    "NEXT_LOWER", "PREV_SPACE", "NEXT_SPACE", "PREV_SYMBOL", "NEXT_AT", "NEXT_SLASH", "AFTER_NEXT_ALPHA",
    "AFTER_NEXT_SLASH",
)

for _bit, _name in enumerate(_FLAG_NAMES):
    globals()[_name + "_MASK"] = 1 << _bit
for _col, _name in enumerate(_COLUMN_NAMES):
    globals()[_name + "_IDX"] = _col
FEATURE_COUNT = len(_COLUMN_NAMES)

del _bit, _col, _name
