"""GPU: the driver's smoke entry point (`__graft_entry__.smoke`) — one small invocation of the hot path against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke(capsys):
    import __graft_entry__ as g
    g.smoke()
    assert 'smoke ok' in capsys.readouterr().out
