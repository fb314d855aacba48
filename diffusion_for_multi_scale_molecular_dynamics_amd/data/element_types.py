"""Element symbols <-> atom-type indices (src/.../data/element_types.py): elements sorted alphabetically are types 0, 1, ...;
one reserved symbol with id -1 stands for padding."""
from typing import List

NULL_ELEMENT = "NULL_ELEMENT_FOR_PADDING"
NULL_ELEMENT_ID = -1


class ElementTypes:
    """Both directions of the map live in ONE tuple: the sorted symbols; index = atom type.  The padding symbol is answered
    beside it."""

    __slots__ = ("_symbols",)

    def __init__(self, elements: List[str]):
        ElementTypes.validate_elements(elements)
        self._symbols = tuple(sorted(elements))

    @staticmethod
    def validate_elements(elements: List[str]):
        assert NULL_ELEMENT not in elements, f"The element '{NULL_ELEMENT}' is reserved and should not be used."
        assert len(set(elements)) == len(elements), "Each entry in the elements list should be unique."

    number_of_atom_types = property(lambda self: len(self._symbols))
    elements = property(lambda self: list(self._symbols))
    element_ids = property(lambda self: list(range(len(self._symbols))))

    def get_element(self, element_id: int) -> str:
        if element_id == NULL_ELEMENT_ID:
            return NULL_ELEMENT
        if not 0 <= element_id < len(self._symbols):
            raise KeyError(element_id)
        return self._symbols[element_id]

    def get_element_id(self, element: str) -> int:
        if element == NULL_ELEMENT:
            return NULL_ELEMENT_ID
        try:
            return self._symbols.index(element)
        except ValueError:
            raise KeyError(element) from None
