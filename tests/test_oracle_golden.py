"""Pins oracle/gridpp_oracle.c against the known-answer values of the
reference's own unit tests (tests/golden/reference_known_answers.json)."""
import pytest

from tests import pins, refapi


@pytest.mark.parametrize("pin", pins.ALL_PINS, ids=lambda f: f.__name__)
def test_oracle_pin(pin, golden):
    pin(refapi, golden)
