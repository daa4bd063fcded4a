"""Text primitives of the generator (same method NAMES as the reference's L1 layer,
helpers/_code_generation_helpers.py:1-33, written afresh on one chunk-list emitter).  Only the vocabulary that survives the redesign is kept:
there is no thread-strided ``gen_add_parallel_loop`` / ``gen_add_sync`` / ``gen_add_serial_ops``
because a configuration is owned by one lane and the emitted ``_inner`` bodies are straight-line.
"""


class TextMixin:
    """Output buffer = a list of text chunks joined on demand; `indent_level` counts 4-space steps.  The method names are the
    reference's (they are what its emitters -- and ours -- are written against); the bodies are not."""
    INDENT = "    "

    def _emit(self, lines, open_scope=False):
        """Append `lines` at the current indentation; open_scope: what follows is nested one level deeper."""
        pad = self.INDENT * self.indent_level
        self._chunks.extend(pad + line + "\n" for line in lines)
        self.indent_level += 1 if open_scope else 0

    def _close_scope(self, blank_after):
        self.indent_level = max(0, self.indent_level - 1)
        self._emit(["}"])
        if blank_after:
            self._chunks.append("\n")

    def gen_add_code_line(self, new_code_line, add_indent_after=False):
        self._emit([new_code_line], add_indent_after)

    def gen_add_code_lines(self, new_code_lines, add_indent_after=False):
        self._emit(list(new_code_lines), add_indent_after)

    def gen_add_raw(self, text):
        """Append pre-indented text verbatim (used for the traced straight-line bodies)."""
        self._chunks.append(text if text.endswith("\n") else text + "\n")

    def gen_add_end_control_flow(self):
        self._close_scope(blank_after=False)

    def gen_add_end_function(self):
        self._close_scope(blank_after=True)

    def gen_add_func_doc(self, func_desc, notes=(), params=(), return_val=None):
        """Doxygen block: description, an optional `Notes:` list, one @param per entry, an optional @return."""
        body = [func_desc, ""]
        if notes:
            body += ["Notes:"] + ["  " + note for note in notes] + [""]
        body += ["@param " + param for param in params]
        if return_val is not None:
            body.append("@return " + return_val)
        self._emit(["/**"] + [(" * " + text).rstrip() for text in body] + [" */"])

    @property
    def code_str(self):
        if len(self._chunks) > 1:
            self._chunks = ["".join(self._chunks)]
        return self._chunks[0] if self._chunks else ""

    @code_str.setter
    def code_str(self, value):
        self._chunks = [value] if value else []

    # column-major static index helpers (same layout rule as the reference)
    @staticmethod
    def gen_static_array_ind_2d(col, row, col_stride=6):
        return col_stride * col + row

    @staticmethod
    def gen_static_array_ind_3d(ind, col, row, ind_stride=36, col_stride=6):
        return ind_stride * ind + col_stride * col + row
