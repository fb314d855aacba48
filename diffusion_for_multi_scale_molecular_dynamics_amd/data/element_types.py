"""Element symbols <-> atom-type indices (src/.../data/element_types.py): elements sorted alphabetically are types 0, 1, ...;
one reserved symbol with id -1 stands for padding."""
from typing import List

NULL_ELEMENT = "NULL_ELEMENT_FOR_PADDING"
NULL_ELEMENT_ID = -1


class ElementTypes:
    def __init__(self, elements: List[str]):
        self.validate_elements(elements)
        self._elements = sorted(elements)
        self._symbol_of = dict(enumerate(self._elements))
        self._symbol_of[NULL_ELEMENT_ID] = NULL_ELEMENT
        self._id_of = {symbol: index for index, symbol in self._symbol_of.items()}

    @staticmethod
    def validate_elements(elements: List[str]):
        assert NULL_ELEMENT not in elements, f"The element '{NULL_ELEMENT}' is reserved and should not be used."
        assert len(set(elements)) == len(elements), "Each entry in the elements list should be unique."

    @property
    def number_of_atom_types(self) -> int:
        return len(self._elements)

    @property
    def elements(self) -> List[str]:
        return self._elements

    @property
    def element_ids(self) -> List[int]:
        return list(range(len(self._elements)))

    def get_element(self, element_id: int) -> str:
        return self._symbol_of[element_id]

    def get_element_id(self, element: str) -> int:
        return self._id_of[element]
