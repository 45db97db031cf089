gfalign search -f testFiles/random3.gfa -n testFiles/random3.search_nodelist.tsv -s 1 -d 4
embedded
1	0	0	-2	2	2	F	1+,4+
2	0	0	-3	3	3	F	1+,2+,4+
3	0	0	-4	4	4	F	1+,2+,3+,4+
