"""paragraph reflow to 118 columns for the design documents (tables, code blocks and list structure kept)"""
import textwrap, re, sys
def reflow(lines, width=118):
    out=[]; code=False; cur=None  # cur = (lead, text)
    def flush():
        nonlocal cur
        if cur is None: return
        lead,text=cur
        sub=' '*len(lead)
        out.extend(textwrap.wrap(text,width=width,initial_indent=lead,subsequent_indent=sub,break_long_words=False,break_on_hyphens=False) or [lead.rstrip()])
        cur=None
    for l in lines:
        if l.startswith('```'):
            flush(); code=not code; out.append(l); continue
        if code: out.append(l); continue
        if not l.strip(): flush(); out.append(''); continue
        if l.lstrip().startswith('|') or l.startswith('#'):
            flush(); out.append(l); continue
        m=re.match(r'^(\s*(?:[*-]|\d+\.)\s+)(.*)$',l)
        if m:
            flush(); cur=(m.group(1),m.group(2).strip()); continue
        if cur is None:
            lead=re.match(r'^\s*',l).group(0); cur=(lead,l.strip())
        else:
            cur=(cur[0],cur[1]+' '+l.strip())
    flush()
    return out
if __name__=='__main__':
    src=open(sys.argv[1]).read().split('\n')
    open(sys.argv[2],'w').write('\n'.join(reflow(src)))
