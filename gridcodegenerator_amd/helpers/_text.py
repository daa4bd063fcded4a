"""Text primitives of the generator (the analogue of the reference's L1 layer,
helpers/_code_generation_helpers.py:1-33).  Only the vocabulary that survives the redesign is kept:
there is no thread-strided ``gen_add_parallel_loop`` / ``gen_add_sync`` / ``gen_add_serial_ops``
because a configuration is owned by one lane and the emitted ``_inner`` bodies are straight-line.
"""


class TextMixin:
    def gen_add_code_line(self, new_code_line, add_indent_after=False):
        self._chunks.append("    " * self.indent_level + new_code_line + "\n")
        if add_indent_after:
            self.indent_level += 1

    def gen_add_code_lines(self, new_code_lines, add_indent_after=False):
        for line in new_code_lines:
            self.gen_add_code_line(line)
        if add_indent_after:
            self.indent_level += 1

    def gen_add_raw(self, text):
        """Append pre-indented text verbatim (used for the traced straight-line bodies)."""
        self._chunks.append(text if text.endswith("\n") else text + "\n")

    def gen_add_end_control_flow(self):
        self.indent_level -= 1
        self.gen_add_code_line("}")

    def gen_add_end_function(self):
        self.indent_level -= 1
        self.gen_add_code_line("}")
        self._chunks.append("\n")

    def gen_add_func_doc(self, func_desc, notes=(), params=(), return_val=None):
        self.gen_add_code_line("/**")
        self.gen_add_code_line(" * " + func_desc)
        self.gen_add_code_line(" *")
        if notes:
            self.gen_add_code_line(" * Notes:")
            for note in notes:
                self.gen_add_code_line(" *   " + note)
            self.gen_add_code_line(" *")
        for param in params:
            self.gen_add_code_line(" * @param " + param)
        if return_val is not None:
            self.gen_add_code_line(" * @return " + return_val)
        self.gen_add_code_line(" */")

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
